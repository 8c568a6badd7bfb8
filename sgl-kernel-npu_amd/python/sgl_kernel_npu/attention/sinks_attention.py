"""Paged attention with sinks and a sliding window, decode and extend forms (reference:
python/sgl_kernel_npu/sgl_kernel_npu/attention/sinks_attention.py:90-137, :241-286).  Both run the paged GQA decode kernel
(csrc/kernels/gqa_decode.hip) with a per-head sink logit in the softmax denominator."""
import torch

import sgl_kernel_npu  # noqa: F401


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def attention_sinks_triton(query, k_cache, v_cache, sinks, block_tables, context_lens, scale, sliding_window_size, q_head_num, k_head_num):
    """Decode: query [S, Hq * D] (one new token per sequence), k_cache / v_cache [blocks, page, Hkv, D], sinks [Hq], block_tables
    [S, max_blocks], context_lens [S]; keys [len - window, len) when sliding_window_size != -1.  Returns [S, Hq * Dv].  (The name is the
    reference's.)"""
    return torch.ops.npu.attention_sinks(query.contiguous(), k_cache, v_cache, sinks.contiguous(), _i32(block_tables), _i32(context_lens).contiguous(),
                                         scale, sliding_window_size, q_head_num, k_head_num, None)


def attention_sinks_prefill_triton(query, k_cache, v_cache, sinks, seq_lens, block_tables, context_lens, scale, sliding_window_size, q_head_num,
                                   k_head_num):
    """Extend: query [sum(seq_lens), Hq * D] = the new tokens of every sequence back to back (seq_lens [B]); context_lens [B] = keys in the cache
    including them.  Token t of sequence b (0 <= t < seq_lens[b]) sees the first context_lens[b] - seq_lens[b] + t + 1 keys (reference
    :168), inside its own sliding window.  No host synchronisation: the per-token tables are formed on the device."""
    S = query.shape[0]
    cum = torch.cumsum(seq_lens.to(torch.int64), dim=0)
    tok = torch.arange(S, device=query.device, dtype=torch.int64)
    b = torch.searchsorted(cum, tok, right=True).clamp_(max=seq_lens.shape[0] - 1)
    kv_len = context_lens.to(torch.int64)[b] - cum[b] + tok + 1
    return torch.ops.npu.attention_sinks(query.contiguous(), k_cache, v_cache, sinks.contiguous(), _i32(block_tables), kv_len.clamp_(min=0).to(torch.int32),
                                         scale, sliding_window_size, q_head_num, k_head_num, b.to(torch.int32))
