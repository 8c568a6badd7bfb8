"""GPT-OSS SwiGLU on interleaved gate / up columns (reference: python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_oai.py:53-104)."""
import torch

import sgl_kernel_npu  # noqa: F401


def swiglu_oai_triton(hidden_states, dim, gemm1_alpha, gemm1_clamp_limit):
    """hidden_states viewed as [-1, dim], gate = even columns, up = odd columns -> [-1, dim / 2] in the input dtype.  (The name is the
    reference's: the kernel is HIP.)"""
    return torch.ops.npu.swiglu_oai(hidden_states.contiguous().view(-1, dim), dim, gemm1_alpha, gemm1_clamp_limit)


def swiglu_oai_native(layer, hidden_states):
    """The reference's torch formulation (:86-96), kept as it is there: arithmetic in the tensors' dtype."""
    E, _, N = layer.w13_weight.size()
    gate_up = hidden_states.view(-1, N)
    alpha = layer.moe_runner_config.gemm1_alpha
    limit = layer.moe_runner_config.gemm1_clamp_limit
    gate, up = gate_up[..., ::2], gate_up[..., 1::2]
    gate = gate.clamp(min=None, max=limit)
    up = up.clamp(min=-limit, max=limit)
    glu = gate * torch.sigmoid(gate * alpha)
    return (up + 1) * glu


def swiglu_oai(layer, hidden_states):
    return swiglu_oai_triton(hidden_states, layer.w13_weight.shape[2], layer.moe_runner_config.gemm1_alpha,
                             layer.moe_runner_config.gemm1_clamp_limit)
