"""out = routed_input * scaling_factor + shared_input: the shared-expert add behind the MoE combine (reference:
python/sgl_kernel_npu/sgl_kernel_npu/moe/mul_add.py:38-60).  The product is rounded to the tensors' dtype before the sum, as the tensor
expression evaluates in that dtype (no reference test: parity unpinned)."""
import torch

import sgl_kernel_npu  # noqa: F401


def mul_add(routed_input, shared_input, scaling_factor):
    return torch.ops.npu.mul_add(routed_input.contiguous(), shared_input.contiguous(), float(scaling_factor))
