"""RMSNorm without a weight (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/rmsnorm_without_weight.py:59-76)."""
import torch

import sgl_kernel_npu  # noqa: F401


def fused_rmsnorm_without_weight(x, eps):
    """x [B, L, C] -> x * rsqrt(mean(x^2, -1) + eps) in x's dtype (bf16 / fp16 / fp32; fp32 arithmetic)."""
    return torch.ops.npu.rmsnorm_without_weight(x.contiguous(), eps)
