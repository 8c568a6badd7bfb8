"""split QKV + per-head RMSNorm + RoPE (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_rope.py:374-438) and its
gated Gemma form (:686-745)."""
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


def split_qkvgate_gemma_rmsnorm_rope(input, sin, cos, q_hidden_size, kv_hidden_size, head_dim, rope_dim, eps, q_weight, k_weight):
    """input [B, 2*q_hidden + 2*kv_hidden] = per q head [q | gate], then K, then V; Gemma RMSNorm (weight + 1) of every q / k head, neox
    RoPE on the first rope_dim dims (sin / cos [B, rope_dim]).  Returns (q [B,q_hidden], k [B,kv_hidden], v [B,kv_hidden], gate [B,q_hidden])."""
    assert head_dim & (head_dim - 1) == 0
    assert q_hidden_size % kv_hidden_size == 0
    return torch.ops.npu.split_qkvgate_gemma_rmsnorm_rope(input, sin.contiguous(), cos.contiguous(), q_hidden_size, kv_hidden_size, head_dim,
                                                          rope_dim, eps, q_weight, k_weight)
