"""Scratch: mla_preprocess at 128 tokens x 128 heads, event-timed per call (p50 / min of 200)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401

H, N, Hh = 7168, int(os.environ.get("MLA_PRE_TOKENS", "128")), 128
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
mode = os.environ.get("MLA_PRE_QUANT", "per_token_quant_symm")
f = lambda: torch.ops.npu.mla_preprocess(hid, gamma0, beta0, wdqkv, descale0, gamma1, beta1, wuq, descale1, gamma2, cos, sin, wuk, kv, kr,
                                         slots, qs0, qo0, bias0, qs1, qo1, bias1, cache_mode="krope_ctkv", quant_mode=mode,
                                         q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
for _ in range(50): f()
torch.cuda.synchronize()
ts = []
for _ in range(200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200): f()
b.record(); torch.cuda.synchronize()
print("mla_preprocess %s us p50 %.1f min %.1f queued %.1f" % (mode, ts[len(ts) // 2], ts[0], a.elapsed_time(b) * 1e3 / 200))
