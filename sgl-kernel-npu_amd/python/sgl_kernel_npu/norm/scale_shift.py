"""x * (c + scale) + shift (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/scale_shift.py:122-183)."""
import torch

import sgl_kernel_npu  # noqa: F401


def fused_scale_shift(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, scale_constant: float = 1.0):
    """x [B, L, C]; scale: one value or C; shift: one value, C, or one per element of x.  c = scale_constant when shift is per element,
    1.0 otherwise (the reference's two kernels).  scale / shift in x's dtype or float32; fp32 arithmetic; returns x's dtype."""
    hidden_size = x.shape[2]
    scale, shift = scale.reshape(-1), shift.reshape(-1)
    assert scale.numel() == 1 or scale.numel() == hidden_size, "scale must be scalar or [hidden_size]"
    assert shift.numel() in (1, hidden_size, x.numel()), "shift must be scalar or [hidden_size]"
    return torch.ops.npu.fused_scale_shift(x.contiguous(), scale.contiguous(), shift.contiguous(), scale_constant)
