"""Paged decode attention entry points (reference: python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py).

decode_mla keeps the reference signature (decode_attention.py:166-175) and in-place `att_out` semantics; the work is
done by the gfx950 kernel behind torch.ops.npu.decode_mla (csrc/kernels/mla_decode.hip).  decode_gqa /
decode_gqa_high_performance (decode_attention.py:378-450, :646-760) run csrc/kernels/gqa_decode.hip (or the MLA kernel
when V is the 512-column prefix view of a 576-wide K cache)."""
import torch

import sgl_kernel_npu  # noqa: F401  (loads the operator library)


def decode_mla(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, sm_scale, page_size, block_table, plan=None):
    """q [B, Hq, 576]; k_nope_buffer [blocks, page, Hkv, 512]; k_rope_buffer [blocks, page, Hkv, 64];
    att_out [B, Hq, 512] (written in place); kv_seq_lens int32 [B]; block_table int32 [B, max_pages].
    plan (MI355X extension, optional): the work list decode_mla_plan(kv_seq_lens) returned -- the attention layers of one decode step
    share it, and the call then does without its own plan launch.  It must come from the same kv_seq_lens contents."""
    if plan is not None:
        torch.ops.npu.decode_mla_planned(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, float(sm_scale), int(page_size),
                                         block_table, plan)
        return att_out
    torch.ops.npu.decode_mla(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, float(sm_scale), int(page_size),
                             block_table, 0)
    return att_out


def decode_mla_plan(kv_seq_lens, num_kv_heads=1):
    """MI355X extension: the length-aware work list of decode_mla for this batch (one small launch, no host sync): how many pieces every
    sequence is cut into so that all pieces fit one round of workgroups, longest first.  Pass it as decode_mla(..., plan=...) to every
    layer of the step."""
    return torch.ops.npu.decode_mla_plan(kv_seq_lens, int(num_kv_heads))


def decode_gqa(q, k_buffer, v_buffer, att_out, kv_seq_lens, sm_scale, page_size, block_table):
    """q [B, Hq, Lk]; k_buffer [blocks, page, Hkv, Lk]; v_buffer [blocks, page, Hkv, Lv] (may be a view of k_buffer);
    att_out [B, Hq, Lv] (written in place); kv_seq_lens int32 [B]; block_table int32 [B, max_pages]."""
    assert q.shape[1] % k_buffer.shape[2] == 0, "head_num must be divisible by kv_head_num"
    torch.ops.npu.decode_gqa(q, k_buffer, v_buffer, att_out, kv_seq_lens, float(sm_scale), int(page_size), block_table, 0)
    return att_out


def decode_gqa_high_performance(q, k_buffer, v_buffer, att_out, kv_seq_lens, qk_out, p_ptr, pv_ptr, sm_scale, page_size,
                                block_table):
    """Same contract as decode_gqa; the reference variant needs caller-provided scratch for its three-pass pipeline
    (qk_out, p_ptr, pv_ptr: decode_attention.py:646-677).  The gfx950 kernel keeps scores in registers, so the scratch
    tensors are accepted and left untouched."""
    return decode_gqa(q, k_buffer, v_buffer, att_out, kv_seq_lens, sm_scale, page_size, block_table)
