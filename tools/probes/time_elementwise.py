"""Scratch: the A11-A13 primitives at the bench_extra shapes, queued back to back (kernel time without the per-call event overhead) and per call."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.activation.swiglu_quant import swiglu_quant
from sgl_kernel_npu.norm.add_rmsnorm_bias import add_rmsnorm_bias
from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkv_rmsnorm_rope

g = torch.Generator(device="cuda").manual_seed(0)


def queued(f, n=100):
    for _ in range(20):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


out = {}
S, h = 32768, 4096
xs = torch.randn((S, h), generator=g, device="cuda").to(torch.bfloat16)
gl = torch.full((32,), S // 32, dtype=torch.int64, device="cuda")
B, H = 4096, 7168
a, r_ = torch.randn((B, H), generator=g, device="cuda").to(torch.bfloat16), torch.randn((B, H), generator=g, device="cuda").to(torch.bfloat16)
wt, bs = torch.randn(H, device="cuda").to(torch.bfloat16), torch.randn(H, device="cuda").to(torch.bfloat16)
qkv = torch.randn((B, 6144 + 2048), generator=g, device="cuda").to(torch.bfloat16)
sn, cs = torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16), torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16)
hw = torch.randn(128, device="cuda").to(torch.bfloat16)
big = torch.randn((4 * B, 6144 + 2048), generator=g, device="cuda").to(torch.bfloat16)
snb, csb = torch.rand((4 * B, 1, 1, 128), device="cuda").to(torch.bfloat16), torch.rand((4 * B, 1, 1, 128), device="cuda").to(torch.bfloat16)
for rnd in range(2):
    t = queued(lambda: swiglu_quant(xs, gl, 1)); out[f"swiglu_quant_{rnd}"] = (round(t, 1), round(S * (h * 2 + h // 2 + 4) / t / 1e3))
    t = queued(lambda: add_rmsnorm_bias(a, r_, wt, bs, 1e-6)); out[f"add_rmsnorm_bias_{rnd}"] = (round(t, 1), round(B * H * 8 / t / 1e3))
    t = queued(lambda: split_qkv_rmsnorm_rope(qkv, sn, cs, 6144, 1024, 128, 1e-6, hw, hw, hw, hw)); out[f"split_qkv_{rnd}"] = (round(t, 1), round(B * 8192 * 4 / t / 1e3))
    t = queued(lambda: split_qkv_rmsnorm_rope(big, snb, csb, 6144, 1024, 128, 1e-6, hw, hw, hw, hw)); out[f"split_qkv_16k_{rnd}"] = (round(t, 1), round(4 * B * 8192 * 4 / t / 1e3))
print(json.dumps(out))
