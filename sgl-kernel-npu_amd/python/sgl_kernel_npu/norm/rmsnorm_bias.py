"""RMSNorm (+bias) (+static INT8 quantisation) without a residual
(reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/rmsnorm_bias.py:8-120)."""
import torch

import sgl_kernel_npu  # noqa: F401


def rmsnorm_bias(input, norm_weight, norm_bias, eps, quant_scale=None, quant_offset=None):
    """-> output [B, H] in the input dtype, or int8 when quant_scale / quant_offset are given."""
    out, _ = torch.ops.npu.add_rmsnorm_bias(input, None, norm_weight, norm_bias, float(eps), quant_scale, quant_offset, False)
    return out
