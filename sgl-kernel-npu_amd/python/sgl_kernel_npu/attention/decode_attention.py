"""Paged decode attention entry points (reference: python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py).

decode_mla keeps the reference signature (decode_attention.py:166-175) and in-place `att_out` semantics; the work is
done by the gfx950 kernel behind torch.ops.npu.decode_mla (csrc/kernels/mla_decode.hip)."""
import torch

import sgl_kernel_npu  # noqa: F401  (loads the operator library)


def decode_mla(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, sm_scale, page_size, block_table):
    """q [B, Hq, 576]; k_nope_buffer [blocks, page, Hkv, 512]; k_rope_buffer [blocks, page, Hkv, 64];
    att_out [B, Hq, 512] (written in place); kv_seq_lens int32 [B]; block_table int32 [B, max_pages]."""
    torch.ops.npu.decode_mla(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, float(sm_scale), int(page_size),
                             block_table, 0)
    return att_out
