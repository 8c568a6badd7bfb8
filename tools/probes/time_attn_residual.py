"""Scratch: kimi_k3 mix_fused at 4096 tokens x 8 bank rows x 7168, event-timed."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.kimi_k3.attn_residual import mix_fused
pf, bk = torch.randn((4096, 7168), device="cuda").bfloat16(), torch.randn((4096, 8, 7168), device="cuda").bfloat16()
cw = torch.randn(7168, device="cuda") * 0.05
for _ in range(3): mix_fused(pf, bk, 8, cw, 1e-6)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): mix_fused(pf, bk, 8, cw, 1e-6)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 20 * 1e3
print(f"mix_fused: {us:.1f} us, {4096 * 7168 * 2 * 10 / us / 1e3:.0f} GB/s algorithmic")
