"""Split QKV + optional RMSNorm + RoPE from a position-indexed cos / sin cache (reference:
python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_rope_pos_cache_half_npu.py:232-407)."""
import torch

import sgl_kernel_npu  # noqa: F401


def split_qkv_rmsnorm_rope_pos_cache_half_npu(input_tensor, positions, cos_sin_cache, q_hidden_size, kv_hidden_size, head_dim, eps=None,
                                              q_weight=None, k_weight=None, q_bias=None, k_bias=None, rope_dim=None, cast_norm_to_bf16=True,
                                              wide_grid_min_tokens=1024):
    """input [B, q_hidden + 2 kv_hidden]; positions [B] (int32 / int64, clamped to the cache inside the kernel: no host sync, graph-safe);
    cos_sin_cache [max_seq, rope_dim] = [cos half | sin half] in any of bf16 / fp16 / fp32.  eps None: no norm.  cast_norm_to_bf16: the
    normalised value is rounded to the I/O dtype before the rotation.  wide_grid_min_tokens is an Ascend launch-shape knob: accepted,
    unused.  Returns (q, k, v).  (The name is the reference's.)"""
    assert input_tensor.dim() == 2
    B, total_hidden = input_tensor.shape
    if rope_dim is None:
        rope_dim = head_dim
    assert rope_dim % 2 == 0 and rope_dim <= head_dim
    assert total_hidden == q_hidden_size + 2 * kv_hidden_size
    pos = positions
    assert pos.numel() == B, f"positions must be [B], got numel={pos.numel()} B={B}"
    if pos.dtype not in (torch.int32, torch.int64):
        pos = pos.to(torch.int32)
    assert head_dim & (head_dim - 1) == 0, "this kernel assumes head_dim is power-of-2"
    assert q_hidden_size % kv_hidden_size == 0
    if eps is not None:
        for name, w in (("q_weight", q_weight), ("k_weight", k_weight)):
            if w is None or w.numel() < head_dim:
                raise ValueError(f"When using RMSNorm (eps is not None), {name} must have at least head_dim={head_dim} elements, "
                                 f"got {w.numel() if w is not None else 0}.")
    if q_bias is not None:
        for name, b in (("q_bias", q_bias), ("k_bias", k_bias)):
            if b is None or b.numel() < head_dim:
                raise ValueError(f"When using bias (q_bias provided), {name} must have at least head_dim={head_dim} elements, "
                                 f"got {b.numel() if b is not None else 0}.")
    return torch.ops.npu.split_qkv_rmsnorm_rope_pos_cache_half(input_tensor.contiguous(), pos.contiguous().reshape(-1), cos_sin_cache.contiguous(),
                                                                q_hidden_size, kv_hidden_size, head_dim, eps, q_weight, k_weight, q_bias, k_bias,
                                                                rope_dim, cast_norm_to_bf16)
