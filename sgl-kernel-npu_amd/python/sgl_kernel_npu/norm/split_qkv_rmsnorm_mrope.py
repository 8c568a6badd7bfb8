"""split QKV (+ gate) + per-head RMSNorm + multimodal RoPE (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_mrope.py:335-471)."""
import torch

import sgl_kernel_npu  # noqa: F401


def triton_split_qkv_rmsnorm_mrope(qkv, q_weight, k_weight, cos_sin, num_q_heads, num_kv_heads, head_size, eps, mrope_section, is_interleaved,
                                   rope_dim=None, q_bias=None, k_bias=None, has_gate=False):
    """qkv [tokens, q (+ gate) + 2 kv] (with has_gate the row starts with num_q_heads pairs [q head | gate head]); cos_sin [3, tokens, rope_dim]
    = (t, h, w) sections, each row [cos half | sin half]; mrope_section = the three section sizes, taken interleaved (o % 3) or contiguous.
    Returns (q, k, v, gate); gate is [tokens, 0] without has_gate.  (The name is the reference's: nothing here is Triton.)"""
    return torch.ops.npu.split_qkv_rmsnorm_mrope(qkv.contiguous(), q_weight, k_weight, cos_sin.contiguous(), num_q_heads, num_kv_heads, head_size, eps,
                                                 list(mrope_section), is_interleaved, rope_dim, q_bias, k_bias, has_gate)


def triton_split_qkv_rmsnorm_mrope_fake(qkv, q_weight, k_weight, cos_sin, num_q_heads, num_kv_heads, head_size, eps, mrope_section, is_interleaved,
                                        rope_dim=None, q_bias=None, k_bias=None, has_gate=False):
    """Shapes and dtypes only (reference :422-471)."""
    T = qkv.shape[0]
    q_size, kv_size = num_q_heads * head_size, num_kv_heads * head_size
    e = lambda n: torch.empty(T, n, device=qkv.device, dtype=qkv.dtype)
    return e(q_size), e(kv_size), e(kv_size), e(q_size if has_gate else 0)
