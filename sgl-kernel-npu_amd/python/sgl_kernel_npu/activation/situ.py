"""SiTU ("SituAndMul") activation with optional MoE group list and per-row int8 (reference:
python/sgl_kernel_npu/sgl_kernel_npu/activation/situ.py:165-480):

    gate' = beta * tanh(gate / beta) * sigmoid(gate);  up' = linear_beta * tanh(up / linear_beta) (linear_beta None: up);  out = gate' * up'
    quantised: scale = max(max|out| / 127, 1e-30), q = clamp(floor(out / scale + 0.5), -128, 127)

on the [gate | up] halves of the last dimension.  The reference holds no test for this file (parity unpinned: the oracle restates the
kernel text).  One deliberate difference: with a cumulative group list (type 0) the reference kernels read the entry BEHIND the list
(`group_list_ptr + NUM_EXPERTS`, :36 / :121 / :381); here the total is the LAST entry, as in swiglu_quant / swiglu_oai_quant."""
from typing import Optional

import torch

import sgl_kernel_npu  # noqa: F401


def _check_group_list(group_list, group_list_type):
    if group_list is not None:
        if group_list_type not in (0, 1):
            raise ValueError(f"group_list_type must be 0 or 1, but got {group_list_type}")
        if group_list.dtype not in (torch.int32, torch.int64):
            raise ValueError(f"group_list dtype must be torch.int32 or torch.int64, but got {group_list.dtype}")


def situ_and_mul(x, group_list=None, group_list_type=None, beta: float = 4.0, linear_beta: Optional[float] = 25.0):
    """x [..., 2d] -> [..., d]; with a group list only the first sum(group_list) rows are written (reference :280-357)."""
    _check_group_list(group_list, group_list_type)
    if x.shape[-1] % 2 != 0:
        raise ValueError(f"x last dim must be even, but got {x.shape[-1]}")
    out, _ = torch.ops.npu.situ_and_mul(x.contiguous(), group_list, group_list_type, float(beta), linear_beta, False)
    return out


def situ_and_mul_quant(x, group_list=None, group_list_type=None, beta: float = 4.0, linear_beta: Optional[float] = 25.0,
                       need_quant: bool = True, quant_type: int = 0):
    """-> (out int8 [..., d], scale fp32 [rows]) for d <= 6144; as the reference (:165-277), fp8 (quant_type 1) and the unquantised /
    large-d form of THIS entry point raise NotImplementedError (use situ_and_mul for the unquantised activation)."""
    if quant_type not in (0, 1):
        raise ValueError(f"quant_type must be 0 (int8) or 1 (fp8), but got {quant_type}")
    if need_quant and quant_type == 1:
        raise NotImplementedError("fp8 (quant_type=1) is deferred in the reference as well; use quant_type=0 (int8).")
    _check_group_list(group_list, group_list_type)
    if x.shape[-1] % 2 != 0:
        raise ValueError(f"x last dim must be even, but got {x.shape[-1]}")
    if not (need_quant and x.shape[-1] // 2 <= 6144):
        raise NotImplementedError("SituAndMul quantization is only implemented for d<=6144 (int8). ")
    return torch.ops.npu.situ_and_mul(x.contiguous(), group_list, group_list_type, float(beta), linear_beta, True)


def situ(hidden_states: torch.Tensor, group_list: torch.Tensor, group_list_type: int, *, need_quant: bool, beta: float = 4.0,
         linear_beta: Optional[float] = 25.0):
    """Grouped Kimi-K3 SiTU with optional INT8 requantisation (reference :429-480) -> (out, scale or None)."""
    if group_list_type not in (0, 1):
        raise ValueError(f"group_list_type must be 0 or 1, got {group_list_type}")
    if hidden_states.ndim != 2 or hidden_states.shape[1] % 2:
        raise ValueError("SiTU input must have shape [tokens, 2 * intermediate]")
    if group_list.dtype not in (torch.int32, torch.int64):
        raise ValueError("group_list must use int32 or int64")
    out, scale = torch.ops.npu.situ_and_mul(hidden_states.contiguous(), group_list, group_list_type, float(beta), linear_beta, bool(need_quant))
    return out, scale if need_quant else None
