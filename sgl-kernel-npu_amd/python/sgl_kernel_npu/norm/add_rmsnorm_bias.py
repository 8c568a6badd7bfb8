"""Fused Add + RMSNorm (+bias) (+static INT8 quantisation) and the Gemma variant
(reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/add_rmsnorm_bias.py:83-147,194-232)."""
import torch

import sgl_kernel_npu  # noqa: F401


def add_rmsnorm_bias(input, residual, norm_weight, norm_bias, eps, quant_scale=None, quant_offset=None):
    """-> (output [B,H] in input dtype, or int8 when quant_scale/quant_offset are given; residual_sum [B,H])."""
    return torch.ops.npu.add_rmsnorm_bias(input, residual, norm_weight, norm_bias, float(eps), quant_scale, quant_offset, False)


def add_gemma_rms_norm(hidden_state, weight, residual, variance_epsilon):
    """-> (norm_output, add_output); x * rsqrt(mean(x^2) + eps) * (w + 1)."""
    return torch.ops.npu.add_rmsnorm_bias(hidden_state, residual, weight, None, float(variance_epsilon), None, None, True)
