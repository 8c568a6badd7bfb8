"""split QKV + per-head RMSNorm + RoPE (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_rope.py:374-438)."""
import torch

import sgl_kernel_npu  # noqa: F401


def split_qkv_rmsnorm_rope(input, sin, cos, q_hidden_size, kv_hidden_size, head_dim, eps=None, q_weight=None, k_weight=None,
                           q_bias=None, k_bias=None, is_neox_style=True):
    """input [B, q_hidden + 2*kv_hidden]; sin/cos [B, 1, 1, rope_dim]; norm skipped when eps is None; partial RoPE when
    rope_dim < head_dim.  Returns (q [B,q_hidden], k [B,kv_hidden], v [B,kv_hidden])."""
    assert head_dim & (head_dim - 1) == 0
    assert q_hidden_size % kv_hidden_size == 0
    return torch.ops.npu.split_qkv_rmsnorm_rope(input, sin.contiguous(), cos.contiguous(), q_hidden_size, kv_hidden_size, head_dim,
                                                eps, q_weight, k_weight, q_bias, k_bias, is_neox_style)
