"""Scratch: low-latency dispatch + combine pairs at BASELINE C3 shape on one rank, 300 pairs (for rocprofv3 kernel traces: prof_any.sh)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29584")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
import deep_ep
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
T, H, K, E = 128, 7168, 8, 32
x = torch.randn((T, H), device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, E), device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), device="cuda")
for _ in range(300):
    (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
    y = rx.to(torch.bfloat16) if rx.dtype != torch.bfloat16 else rx
    out, _, _ = buf.low_latency_combine(y, idx, w, handle)
torch.cuda.synchronize()
