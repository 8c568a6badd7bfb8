"""Scratch: kernel-only time of split_qkv_rmsnorm_rope / add_rmsnorm_bias at 4096 rows (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkv_rmsnorm_rope

B = 4096
qkv = torch.randn((B, 6144 + 2048), device="cuda").to(torch.bfloat16)
sn, cs = torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16), torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16)
hw = torch.randn(128, device="cuda").to(torch.bfloat16)
for _ in range(50):
    split_qkv_rmsnorm_rope(qkv, sn, cs, 6144, 1024, 128, 1e-6, hw, hw, hw, hw)
torch.cuda.synchronize()
