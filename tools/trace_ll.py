"""Scratch: kernel timeline of one low-latency dispatch + combine (run under rocprofv3 --kernel-trace, then parse)."""
import csv, os, sys
if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "combine_reduce" in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["End_Timestamp"]); prev = t0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:80]}")
        prev = e
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("gloo", rank=0, world_size=1)
import deep_ep
H, K, E, T = 7168, 8, 32, 128
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), generator=g, device="cuda")
for _ in range(20):
    (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
    y = (rx.float() * rs[:, None]).to(torch.bfloat16)
    buf.low_latency_combine(y, idx, w, handle)
torch.cuda.synchronize()
