"""Scratch: 60 low-latency dispatch + combine calls at 128 tokens (run under rocprofv3 --kernel-trace; then gap_analysis-style print)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import deep_ep
H, K, E, T = 7168, 8, 32, 128
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), generator=g, device="cuda")
(rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
y = (rx.float() * rs[:, None]).to(torch.bfloat16)
for _ in range(60):
    (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
    buf.low_latency_combine(y, idx, w, handle)
torch.cuda.synchronize()
dist.destroy_process_group()
