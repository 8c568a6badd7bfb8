"""Scratch: SiTU (quantised and not) and swiglu_oai_quant at 16384 rows, event-timed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.activation.situ import situ_and_mul, situ_and_mul_quant
from sgl_kernel_npu.activation.swiglu_oai_quant import swiglu_oai_quant
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for d in (3072, 6144):
    x = torch.randn(16384, 2 * d, device="cuda").to(torch.bfloat16)
    us = t(lambda: situ_and_mul_quant(x)); print(f"situ_and_mul_quant d={d}: {us:.1f} us, {16384 * (4 * d + d + 4) / us / 1e3:.0f} GB/s")
    us = t(lambda: situ_and_mul(x)); print(f"situ_and_mul d={d}: {us:.1f} us, {16384 * (4 * d + 2 * d) / us / 1e3:.0f} GB/s")
x = torch.randn(16384, 5760, device="cuda").to(torch.bfloat16)
us = t(lambda: swiglu_oai_quant(x, 1.702, 7.0)); print(f"swiglu_oai_quant 2880: {us:.1f} us")
