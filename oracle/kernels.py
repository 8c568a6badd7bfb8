"""CPU restatements of the fused inference primitives (MLA decode, SwiGLU-quant, Add+RMSNorm, split-QKV RMSNorm+RoPE).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  torch on CPU is used as the array library (it has bf16/fp16);
every function cites the reference lines it follows."""
import math

import numpy as np
import torch


# --------------------------------------------------------------------------------------
# A9  paged MLA decode
# --------------------------------------------------------------------------------------
def decode_mla(q, k_nope, k_rope, kv_seq_lens, block_table, sm_scale, page_size=None):
    """Restates _paged_mla_fwd_kernel (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:110-163):
    per page: S = (q_nope·K_nope^T + q_pe·K_rope^T) * sm_scale with fp32 accumulation of low-precision products,
    mask beyond kv_seq_lens, online softmax (m, l fp32), P cast to the KV dtype, acc += P·K_nope (V aliases K_nope, :123),
    out = acc / l.  q [B,Hq,Dn+Dr]; k_nope [blocks,page,Hkv,Dn]; k_rope [blocks,page,Hkv,Dr]; returns [B,Hq,Dn] in q.dtype.
    """
    B, Hq, D = q.shape
    nb, page, Hkv, Dn = k_nope.shape
    Dr = k_rope.shape[-1]
    assert D == Dn + Dr
    group = Hq // Hkv
    out = torch.zeros((B, Hq, Dn), dtype=q.dtype)
    qf = q.float()
    for b in range(B):
        L = int(kv_seq_lens[b])
        npages = (L + page - 1) // page
        for kvh in range(Hkv):
            hs = slice(kvh * group, (kvh + 1) * group)
            qn, qr = qf[b, hs, :Dn], qf[b, hs, Dn:]
            m = torch.full((group,), -float("inf"))
            l = torch.zeros(group)
            acc = torch.zeros((group, Dn))
            for pg in range(npages):
                blk = int(block_table[b, pg])
                kn = k_nope[blk, :, kvh, :].float()
                kr = k_rope[blk, :, kvh, :].float()
                s = (qn @ kn.T + qr @ kr.T) * sm_scale
                valid = (pg * page + torch.arange(page)) < L
                s = torch.where(valid[None, :], s, torch.tensor(-float("inf")))
                m_new = torch.maximum(s.max(dim=1).values, m)
                alpha = torch.exp(m - m_new)
                p = torch.exp(s - m_new[:, None])
                l = l * alpha + p.sum(dim=1)
                acc = acc * alpha[:, None] + p.to(q.dtype).float() @ kn
                m = m_new
            out[b, hs] = (acc / l[:, None]).to(q.dtype)
    return out


def decode_mla_golden(q, k_nope, k_rope, kv_seq_lens, block_table, sm_scale):
    """Transcription of the reference TEST golden decode_mla_golden
    (tests/python/sgl_kernel_npu/test_decode_attention.py:131-187): gather the pages, one-shot softmax in fp32,
    scores cast to the value dtype, einsum with V = K_nope."""
    B, Hq, D = q.shape
    nb, page, Hkv, Dn = k_nope.shape
    rep = Hq // Hkv
    outs = []
    for b in range(B):
        L = int(kv_seq_lens[b])
        npages = (L + page - 1) // page
        idx = block_table[b, :npages].long()
        kn = k_nope[idx].reshape(-1, Hkv, Dn)[:L]
        kr = k_rope[idx].reshape(-1, Hkv, k_rope.shape[-1])[:L]
        if rep != 1:
            kn = torch.repeat_interleave(kn, rep, dim=1)
            kr = torch.repeat_interleave(kr, rep, dim=1)
        qq = q[b:b + 1]
        qk = (torch.einsum("qhd,khd->hqk", qq[:, :, :Dn], kn).float() + torch.einsum("qhd,khd->hqk", qq[:, :, Dn:], kr).float()) * sm_scale
        score = torch.softmax(qk, dim=-1).to(kn.dtype)
        outs.append(torch.einsum("hqk,khd->qhd", score, kn))
    return torch.cat(outs, dim=0)
