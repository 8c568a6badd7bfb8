"""Sparse + causal prefill attention with a per-query block table (reference:
python/sgl_kernel_npu/sgl_kernel_npu/attention/fia_blockq_attention.py:89-181).

Every query token attends to its top-k selected KV blocks plus its own block (causally: up to its own position).  The per-query
preparation of the reference's `_fia_prep_kernel` (:11-88: own block last, logical blocks -> physical pages, actual_kvlen) is one HIP
launch; the attention is the paged GQA decode kernel with one query row per "sequence" -- the reference hands the same tables to
`npu_fused_infer_attention_score` after a `.tolist()` of the lengths (a host round trip this path does not make).  No reference test
exists for this file (parity unpinned; tests compare with a restatement of the prep kernel and an fp32 softmax attention)."""
from __future__ import annotations

from typing import Optional

import torch

import sgl_kernel_npu  # noqa: F401


def flash_prefill_bnsd_blockq_sparse_fia(q: torch.Tensor, k_cache_bnsd: torch.Tensor, v_cache_bnsd: torch.Tensor, topk_idx: torch.Tensor,
                                         seq_lens: torch.Tensor, per_query_req: torch.Tensor, req_to_token: torch.Tensor, block_size: int,
                                         sm_scale: Optional[float], num_pages: int, max_num_blocks: int,
                                         block_table_out: Optional[torch.Tensor] = None,
                                         actual_kvlen_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """q [total_q, Hq, D]; caches [num_pages, block_size, 1, D]; topk_idx [1, total_q, topk + 1] int32 (-1 pads); seq_lens [total_q]
    int32 (position + 1); per_query_req [total_q]; req_to_token [requests, max_ctx] int32.  -> [total_q, Hq, D]."""
    assert q.dtype in (torch.float16, torch.bfloat16)
    total_q, num_q_heads, head_dim = q.shape
    num_pages_c, block_size_c, num_kv_heads, cache_head_dim = k_cache_bnsd.shape
    assert block_size_c == block_size
    assert cache_head_dim == head_dim
    assert k_cache_bnsd.shape == v_cache_bnsd.shape
    assert num_kv_heads == 1, f"FIA sparse path supports num_kv_heads==1; got {num_kv_heads} -- use the triton blockq path instead."
    assert topk_idx.shape[0] == num_kv_heads
    assert topk_idx.shape[1] == total_q
    assert topk_idx.dtype == torch.int32
    if sm_scale is None:
        sm_scale = head_dim ** -0.5
    return torch.ops.npu.fia_blockq_sparse_prefill(q, k_cache_bnsd, v_cache_bnsd, topk_idx[0], seq_lens, per_query_req, req_to_token, block_size,
                                                   float(sm_scale), block_table_out, actual_kvlen_out)
