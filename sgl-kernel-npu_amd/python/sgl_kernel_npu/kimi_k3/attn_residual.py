"""Kimi-K3 attention-residual score + combine (reference: python/sgl_kernel_npu/sgl_kernel_npu/kimi_k3/attn_residual.py:66-111): per
token, the first `num_valid_blocks` rows of the residual bank and the prefix row are scored (sum(rmsnorm(row) * combined_weight)),
soft-maxed, and the output is the probability-weighted sum of those rows.  One HIP launch, a wave per token; no reference test exists
(parity unpinned)."""
import torch

import sgl_kernel_npu  # noqa: F401


def mix_fused(prefix_sum: torch.Tensor, bank: torch.Tensor, num_valid_blocks: int, combined_weight: torch.Tensor,
              variance_epsilon: float) -> torch.Tensor:
    num_tokens, _ = prefix_sum.shape
    if num_tokens == 0:
        return prefix_sum
    if not 0 <= num_valid_blocks <= bank.shape[1]:
        raise ValueError("num_valid_blocks must fit within the residual bank")
    return torch.ops.npu.attn_residual_mix(prefix_sum, bank, int(num_valid_blocks), combined_weight.contiguous(), float(variance_epsilon))
