"""Scratch: fused_deep_moe at BASELINE C5 (4096 tokens, 32 local experts) -- p50 of 30 queued calls, three repetitions; for env A/B loops."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29589")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
import deep_ep
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
H, I, K, E, T = 7168, 2048, 8, 32, 4096
g = torch.Generator(device="cuda").manual_seed(0)
w13 = torch.randint(-127, 128, (E, 2 * I, H), generator=g, device="cuda", dtype=torch.int8)
w2 = torch.randint(-127, 128, (E, H, I), generator=g, device="cuda", dtype=torch.int8)
s13 = torch.rand((E, 2 * I), generator=g, device="cuda") * 4e-5 + 1.5e-4
s2 = torch.rand((E, H), generator=g, device="cuda") * 4e-5 + 1.5e-4
x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), generator=g, device="cuda")
f = lambda: buf.fused_deep_moe(x, idx, w, w13, s13, w2, s2, T, E)
for _ in range(10): f()
torch.cuda.synchronize()
ts = []
for rep in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): f()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 30 * 1e3)
print(sys.argv[1] if len(sys.argv) > 1 else "", " ".join(f"{t:.1f}" for t in ts), "us", flush=True)
