"""SwiGLU + per-row INT8 quantisation (reference: python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_quant.py:87-127)."""
import torch

import sgl_kernel_npu  # noqa: F401


def swiglu_quant(x, group_list, group_list_type, need_quant=True, do_limit=False, limit=7.0):
    """x [s, 2I] bf16/fp16 grouped rows; group_list [E] int32/int64 (type 0 = cumulative, 1 = counts).
    Returns (out [s, I] int8 (or x.dtype when need_quant is False), scale [s] f32); rows beyond the group total are
    left uninitialised, as in the reference."""
    if group_list_type not in [0, 1]:
        raise ValueError(f"group_list_type must be 0 or 1, but got {group_list_type}")
    if group_list.dtype not in (torch.int32, torch.int64):
        raise ValueError(f"group_list dtype must be torch.int32 or torch.int64, but got {group_list.dtype}")
    return torch.ops.npu.swiglu_quant(x, group_list, group_list_type, need_quant, do_limit, float(limit))
