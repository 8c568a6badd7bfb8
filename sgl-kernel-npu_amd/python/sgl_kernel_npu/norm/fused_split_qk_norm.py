"""Split of the MLA down-projection + the two RMSNorms (reference: python/sgl_kernel_npu/sgl_kernel_npu/norm/fused_split_qk_norm.py:93-134)."""
import torch

import sgl_kernel_npu  # noqa: F401


def fused_split_qk_norm(fused_qkv_a_proj_out, q_a_layernorm, kv_a_layernorm, q_lora_rank, kv_lora_rank, qk_rope_dim, eps=1e-6):
    """fused_qkv_a_proj_out [B, q_lora_rank + kv_lora_rank + qk_rope_dim]; the two layer-norm modules supply .weight and, if they have one,
    .bias.  Returns (q_lora [B, q_lora_rank], k_nope [B, 1, kv_lora_rank], k_pe [B, 1, qk_rope_dim])."""
    assert q_lora_rank > 0, f"q_lora_rank should be positive, got {q_lora_rank}"
    assert kv_lora_rank > 0, f"kv_lora_rank should be positive, got {kv_lora_rank}"
    assert qk_rope_dim > 0, f"qk_rope_dim should be positive, got {qk_rope_dim}"
    qb = getattr(q_a_layernorm, "bias", None)
    kb = getattr(kv_a_layernorm, "bias", None)
    q_lora, k_nope, k_pe = torch.ops.npu.fused_split_qk_norm(fused_qkv_a_proj_out.contiguous(), q_a_layernorm.weight.contiguous(),
                                                             None if qb is None else qb.contiguous(), kv_a_layernorm.weight.contiguous(),
                                                             None if kb is None else kb.contiguous(), q_lora_rank, kv_lora_rank, qk_rope_dim, eps)
    return q_lora, k_nope.unsqueeze(1), k_pe.unsqueeze(1)
