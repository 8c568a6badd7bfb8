"""L1 normalisation of rows (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/l1_norm.py:29-38)."""
import torch

import sgl_kernel_npu  # noqa: F401


def l1_norm(input):
    """input [batch, hidden] (bf16 / fp16 / fp32) -> fp32 [batch, hidden] = input / sum(input, -1)."""
    return torch.ops.npu.l1_norm(input.contiguous())
