"""split QKV + tensor-parallel RMSNorm + RoPE (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_tp_rmsnorm_rope.py:179-288)."""
import torch
import torch.distributed as dist

import sgl_kernel_npu  # noqa: F401


def split_qkv_tp_rmsnorm_rope(input, cos, sin, q_hidden_size, kv_hidden_size, head_dim, eps, q_weight, k_weight, rotary_dim, tp_world, tp_group):
    """input [B, q_hidden + 2 kv_hidden] = this rank's shard of [Q | K | V]; the RMSNorm of Q and of K runs over the whole (global) hidden
    dimension: the local mean of squares is all-reduced over tp_group between the two launches and multiplied by 1 / tp_world; weights
    [q_hidden], [kv_hidden]; neox RoPE on the first rotary_dim dims of every head with the first half of cos / sin [B, rotary_dim].
    Returns (q [B, q_hidden], k [B, kv_hidden], v [B, kv_hidden])."""
    assert head_dim & (head_dim - 1) == 0
    assert q_hidden_size % kv_hidden_size == 0
    B = input.shape[0]
    if B == 0:
        e = lambda n: torch.empty(0, n, device=input.device, dtype=input.dtype)
        return e(q_hidden_size), e(kv_hidden_size), e(kv_hidden_size)
    input = input.contiguous()
    v, qk_var = torch.ops.npu.split_qkv_tp_local_var(input, q_hidden_size, kv_hidden_size)
    if tp_world > 1:
        dist.all_reduce(qk_var, group=tp_group)
    q, k = torch.ops.npu.split_qkv_tp_norm_rope(input, cos.contiguous(), sin.contiguous(), qk_var, q_hidden_size, kv_hidden_size, head_dim, eps,
                                                q_weight.contiguous(), k_weight.contiguous(), rotary_dim, 1.0 / tp_world)
    return q, k, v
