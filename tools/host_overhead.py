"""Scratch: host wall time vs device time per call (is the Python / C++ host path the bottleneck at decode sizes?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
import deep_ep
H, K, E, T = 7168, 8, 32, 128
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), generator=g, device="cuda")
(rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
y = (rx.float() * rs[:, None]).to(torch.bfloat16)
def loop(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t_issue = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / n * 1e6
    return round(t_issue, 1), round(t_total, 1)
print("LL dispatch  host-issue us / wall us per call:", loop(lambda: buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)))
print("LL combine   host-issue us / wall us per call:", loop(lambda: buf.low_latency_combine(y, idx, w, handle)))
