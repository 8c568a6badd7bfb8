"""Scratch: kernel timeline of one mla_preprocess call (run under rocprofv3 --kernel-trace; pass the csv to print it)."""
import csv, os, sys
if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "pre_quant" in r["Kernel_Name"]]
    a, b = idx[-3], idx[-2]
    t0 = int(rows[a]["End_Timestamp"]); prev = t0
    for r in rows[a + 1:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
        prev = e
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sgl_kernel_npu
H = 7168
# ---- mla_preprocess (decode: 128 tokens, DeepSeek-V3 shapes: hidden 7168, 128 heads)
N, Hh = 128, 128
dt = torch.bfloat16
dd = dict(device="cuda")
hid = (torch.randn(N, H, **dd) * 0.5).to(dt)
wdqkv = torch.randint(-8, 8, (2112, H), dtype=torch.int8, **dd)
wuq = torch.randint(-8, 8, (Hh * 192, 1536), dtype=torch.int8, **dd)
descale0, descale1 = torch.rand(2112, **dd) * 1e-3 + 5e-4, torch.rand(Hh * 192, **dd) * 1e-3 + 5e-4
bias0, bias1 = torch.randint(-50, 50, (2112,), dtype=torch.int32, **dd), torch.randint(-50, 50, (Hh * 192,), dtype=torch.int32, **dd)
gamma0, beta0 = torch.randn(H, **dd).to(dt), torch.randn(H, **dd).to(dt)
gamma1, beta1, gamma2 = torch.randn(1536, **dd).to(dt), torch.randn(1536, **dd).to(dt), torch.randn(512, **dd).to(dt)
wuk = (torch.randn(Hh, 128, 512, **dd) * 0.1).to(dt)
cos, sin = torch.rand(N, 64, **dd).to(dt), torch.rand(N, 64, **dd).to(dt)
qs0, qo0 = torch.tensor([0.02], **dd).to(dt), torch.tensor([3], dtype=torch.int8, **dd)
qs1, qo1 = torch.tensor([0.03], **dd).to(dt), torch.tensor([-2], dtype=torch.int8, **dd)
slots = torch.randperm(4096, **dd)[:N].to(torch.int32)
kv, kr = torch.zeros((32, 128, 1, 512), dtype=dt, **dd), torch.zeros((32, 128, 1, 64), dtype=dt, **dd)
q0, q1 = torch.empty((N, Hh, 512), dtype=dt, **dd), torch.empty((N, Hh, 64), dtype=dt, **dd)
f = lambda: torch.ops.npu.mla_preprocess(hid, gamma0, beta0, wdqkv, descale0, gamma1, beta1, wuq, descale1, gamma2, cos, sin, wuk, kv, kr,
                                         slots, qs0, qo0, bias0, qs1, qo1, bias1, cache_mode="krope_ctkv",
                                         quant_mode="per_tensor_quant_asymm", q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
for _ in range(10): f()
torch.cuda.synchronize()
