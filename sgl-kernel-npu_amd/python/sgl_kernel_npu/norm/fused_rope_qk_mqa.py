"""RoPE on q and the shared key heads (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/fused_rope_qk_mqa.py:113-160)."""
import torch

import sgl_kernel_npu  # noqa: F401


def fused_rope_qk_mqa(query, key, cos_sin, rotary_dim, is_neox_style):
    """query [T, Hq, D], key [T, Hk, D], cos_sin [T, rotary_dim] (cos | sin halves, one row per token).
    Returns (out_q, out_k): the first rotary_dim dims rotated, the rest copied."""
    return torch.ops.npu.fused_rope_qk_mqa(query, key, cos_sin, int(rotary_dim), bool(is_neox_style))
