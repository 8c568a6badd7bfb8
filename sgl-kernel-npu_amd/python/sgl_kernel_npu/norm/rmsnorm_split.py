"""The two halves of a split RMSNorm (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/rmsnorm_split.py:76-97, :150-161): the row
variance on its own (so that it can be all-reduced across tensor-parallel ranks), then x * rsqrt(variance + eps) * weight."""
import torch

import sgl_kernel_npu  # noqa: F401


def fused_rsqrt_mul(x, variance, weight, eps=1e-6):
    """x [B, L, C], variance [B * L] (any shape with that many values), weight [C], all in x's dtype -> [B, L, C]."""
    return torch.ops.npu.fused_rsqrt_mul(x.contiguous(), variance.contiguous().reshape(-1), weight.contiguous(), eps)


def fused_variance(x: torch.Tensor):
    """x [B, L, C] -> mean(x^2, -1) as [B, L, 1] in x's dtype."""
    return torch.ops.npu.fused_variance(x.contiguous())
