""""Zero experts" of type identity (reference: python/sgl_kernel_npu/sgl_kernel_npu/moe/zero_experts_compute_identity.py:50-81): the
selections of a token that point past the real experts contribute `hidden * (sum of their scales)`; IN PLACE their scales become 0 and
their indices `identity_mask_value` (the first one 0 when all K selections of the token were zero experts).  No reference test: parity
unpinned."""
import torch

import sgl_kernel_npu  # noqa: F401


def zero_experts_compute_identity_triton(expert_indices, expert_scales, num_experts, zero_expert_type, hidden_states, identity_mask_value=0):
    """-> zero_result [S, D]; expert_indices / expert_scales [S, K] are modified in place (they must be contiguous)."""
    return torch.ops.npu.zero_experts_compute_identity(expert_indices, expert_scales, int(num_experts), hidden_states.contiguous(),
                                                       int(identity_mask_value))
