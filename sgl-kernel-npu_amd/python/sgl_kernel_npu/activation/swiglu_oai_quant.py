"""swiglu_oai on concatenated [gate | up] halves with optional per-row int8 quantisation (reference:
python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_oai_quant.py:115-211).

    gate = x1.clamp(-inf, limit);  up = x2.clamp(-limit, limit);  out = gate * sigmoid(gate * alpha) * (up + 1)

Quantised: scale = max|out| / 127 per row, q = int8(out / scale) -- rounded to the input dtype first and then TRUNCATED toward zero with
saturation: the reference leaves the conversion to its backend's cast; truncation is the Triton language's float -> int conversion.  No
reference test or vector exists for this function (parity unpinned)."""
import torch

import sgl_kernel_npu  # noqa: F401


def swiglu_oai_quant(x, alpha, limit, need_quant=True, group_list=None, group_list_type=None):
    """x [..., 2d].  Dense mode (group_list None): every row.  MoE grouped mode: group_list = per-expert token counts (type 1) or their
    cumulative sums (type 0); rows past the total are left uninitialised.  Returns (out [..., d] int8 or x.dtype, scale [num_rows] fp32)."""
    if group_list is not None:
        if group_list_type not in (0, 1):
            raise ValueError(f"group_list_type must be 0 or 1, got {group_list_type}")
        if group_list.dtype not in (torch.int32, torch.int64):
            raise ValueError(f"group_list dtype must be torch.int32 or torch.int64, got {group_list.dtype}")
    return torch.ops.npu.swiglu_oai_quant(x.contiguous(), alpha, limit, need_quant, group_list, group_list_type)
