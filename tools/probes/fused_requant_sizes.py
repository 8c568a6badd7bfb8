"""Scratch: fused_deep_moe with the requantisation in GEMM1's epilogue on / off (buf.runtime.set_fused_requant), one process, alternating, at
decode and prefill sizes (32 local experts, DeepSeek-V3 shapes)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29587")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
import deep_ep
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
H, I, K, E = 7168, 2048, 8, 32
g = torch.Generator(device="cuda").manual_seed(0)
w13 = torch.randint(-16, 16, (E, 2 * I, H), generator=g, device="cuda", dtype=torch.int8)
w2 = torch.randint(-16, 16, (E, H, I), generator=g, device="cuda", dtype=torch.int8)
s13 = torch.rand((E, 2 * I), generator=g, device="cuda") * 4e-4 + 1.5e-3
s2 = torch.rand((E, H), generator=g, device="cuda") * 4e-4 + 1.5e-3
for T in (16, 128, 512, 1024, 4096):
    x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
    idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
    w = torch.rand((T, K), generator=g, device="cuda")
    f = lambda: buf.fused_deep_moe(x, idx, w, w13, s13, w2, s2, T, E)
    outs, res = {}, {0: [], 1: []}
    for rep in range(3):
        for on in (0, 1):
            buf.runtime.set_fused_requant(bool(on))
            for _ in range(5): o = f()
            torch.cuda.synchronize()
            outs[on] = o[0].clone()
            n = 50 if T <= 1024 else 15
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): f()
            b.record(); torch.cuda.synchronize()
            res[on].append(a.elapsed_time(b) / n * 1e3)
    print(f"T={T}: rowquant launch {min(res[0]):.1f} us | in GEMM1's epilogue {min(res[1]):.1f} us | identical outputs {bool(torch.equal(outs[0], outs[1]))}", flush=True)
