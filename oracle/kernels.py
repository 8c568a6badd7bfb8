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


# --------------------------------------------------------------------------------------
# A10  paged GQA decode (separate V cache)
# --------------------------------------------------------------------------------------
def decode_gqa(q, k_buffer, v_buffer, kv_seq_lens, block_table, sm_scale):
    """Restates _paged_gqa_fwd_kernel (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:292-375):
    per page S = (q.K^T) * sm_scale (fp32 accumulation of low-precision products), mask beyond kv_seq_lens, online
    softmax, P cast to the V dtype (:362), acc += P.V, out = acc / l.  q [B,Hq,Lk]; k_buffer [blocks,page,Hkv,Lk];
    v_buffer [blocks,page,Hkv,Lv] (may be a view of k_buffer); returns [B,Hq,Lv] in q.dtype."""
    B, Hq, Lk = q.shape
    nb, page, Hkv, _ = k_buffer.shape
    Lv = v_buffer.shape[-1]
    group = Hq // Hkv
    out = torch.zeros((B, Hq, Lv), dtype=q.dtype)
    qf = q.float()
    for b in range(B):
        L = int(kv_seq_lens[b])
        npages = (L + page - 1) // page
        for kvh in range(Hkv):
            hs = slice(kvh * group, (kvh + 1) * group)
            m = torch.full((group,), -float("inf"))
            l = torch.zeros(group)
            acc = torch.zeros((group, Lv))
            for pg in range(npages):
                blk = int(block_table[b, pg])
                k = k_buffer[blk, :, kvh, :].float()
                v = v_buffer[blk, :, kvh, :].float()
                s = (qf[b, hs] @ k.T) * sm_scale
                valid = (pg * page + torch.arange(page)) < L
                s = torch.where(valid[None, :], s, torch.tensor(-float("inf")))
                m_new = torch.maximum(s.max(dim=1).values, m)
                alpha = torch.exp(m - m_new)
                p = torch.exp(s - m_new[:, None])
                l = l * alpha + p.sum(dim=1)
                acc = acc * alpha[:, None] + p.to(q.dtype).float() @ v
                m = m_new
            out[b, hs] = (acc / l[:, None]).to(q.dtype)
    return out


def decode_gqa_golden(q, k_buffer, v_buffer, kv_seq_lens, block_table, sm_scale):
    """Transcription of the reference TEST golden decode_gqa_golden (tests/python/sgl_kernel_npu/test_decode_attention.py:18-60):
    q scaled in its own dtype first (`q *= scale`, :40), one-shot fp32 softmax, scores cast to the V dtype, einsum with V."""
    B, Hq, Lk = q.shape
    nb, page, Hkv, _ = k_buffer.shape
    Lv = v_buffer.shape[-1]
    rep = Hq // Hkv
    outs = []
    for b in range(B):
        L = int(kv_seq_lens[b])
        npages = (L + page - 1) // page
        idx = block_table[b, :npages].long()
        k = k_buffer[idx].reshape(-1, Hkv, Lk)[:L]
        v = v_buffer[idx].reshape(-1, Hkv, Lv)[:L]
        if rep != 1:
            k = torch.repeat_interleave(k, rep, dim=1)
            v = torch.repeat_interleave(v, rep, dim=1)
        qq = q[b:b + 1] * sm_scale
        qk = torch.einsum("qhd,khd->hqk", qq, k).float()
        score = torch.softmax(qk, dim=-1).to(v.dtype)
        outs.append(torch.einsum("hqk,khd->qhd", score, v))
    return torch.cat(outs, dim=0)


# --------------------------------------------------------------------------------------
# A11  SwiGLU + per-row INT8 quantisation
# --------------------------------------------------------------------------------------
def swiglu_quant(x, group_list, group_list_type, need_quant=True, do_limit=False, limit=7.0):
    """Restates _swiglu_quant_kernel (python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_quant.py:26-84):
    rows < total only (total = last cumulative entry for type 0 -- the reference indexes one past the end there, :27 --
    or sum of counts for type 1); out = x1*sigmoid(x1)*x2 in fp32 (optional clamps :52-56); scale = max|out|/127;
    q = clamp(floor(out/scale + 0.5), -128, 127) (:62-72).  Rows >= total are left zero here (uninitialised there).
    Returns (out int8/ dtype [S, I], scale f32 [S], total)."""
    S, h = x.shape
    I = h // 2
    gl = group_list.to(torch.int64)
    total = int(gl[-1]) if group_list_type == 0 else int(gl.sum())
    total = min(total, S)
    xf = x[:total].float()
    x1, x2 = xf[:, :I], xf[:, I:]
    gate = x1 * torch.sigmoid(x1)
    if do_limit:
        gate = torch.minimum(gate, torch.tensor(float(limit)))
        up = torch.clamp(x2, -limit, limit)
        o = gate * up
    else:
        o = gate * x2
    scale = torch.zeros(S, dtype=torch.float32)
    if not need_quant:
        out = torch.zeros((S, I), dtype=x.dtype)
        out[:total] = o.to(x.dtype)
        return out, scale, total
    sc = o.abs().amax(dim=1) / 127.0
    scale[:total] = sc
    q = torch.floor(o / sc[:, None] + 0.5).clamp(-128, 127)
    q = torch.where(sc[:, None] > 0, q, torch.zeros_like(q))
    out = torch.zeros((S, I), dtype=torch.int8)
    out[:total] = q.to(torch.int8)
    return out, scale, total


# --------------------------------------------------------------------------------------
# A12  Add + RMSNorm (+bias) (+static quant) and the Gemma variant
# --------------------------------------------------------------------------------------
def add_rmsnorm_bias(input, residual, norm_weight, norm_bias, eps, quant_scale=None, quant_offset=None, gemma=False):
    """Restates add_rmsnorm_bias_kernel (python/sgl_kernel_npu/sgl_kernel_npu/norm/add_rmsnorm_bias.py:25-77): the sum is
    formed and stored in the input dtype (:33-37), then fp32: y * (1/sqrt(mean(y^2)+eps)) * w + b (:39-47), optional
    int8_saturate(v*quant_scale + quant_offset) (:56-68; rounding = nearest-even, the test golden uses np.round with a
    +-1 tolerance).  gemma=True: add_gemma_rms_norm_kernel (:173-190): rsqrt, weight + 1, no bias."""
    y = input if residual is None else (input + residual)          # dtype add == round(float add)
    yf = y.float()
    var = (yf * yf).mean(dim=-1, keepdim=True)
    rstd = torch.rsqrt(var + eps) if gemma else 1.0 / torch.sqrt(var + eps)
    w = norm_weight.float() + 1.0 if gemma else norm_weight.float()
    v = (yf * rstd) * w
    if norm_bias is not None:
        v = v + norm_bias.float()
    if quant_scale is not None:
        v = torch.round(v * quant_scale.float() + quant_offset.float()).clamp(-128, 127).to(torch.int8)
    else:
        v = v.to(input.dtype)
    return v, y


# --------------------------------------------------------------------------------------
# A13  split QKV + per-head RMSNorm + RoPE
# --------------------------------------------------------------------------------------
def split_qkv_rmsnorm_rope(qkv, sin, cos, q_hidden, kv_hidden, head_dim, eps=None, q_weight=None, k_weight=None, q_bias=None,
                           k_bias=None, is_neox_style=True):
    """Restates split_qkv_rmsnorm_rope_kernel (python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_rope.py:38-198)
    in fp32: per head optional RMSNorm*w(+b) (:55-69), RoPE on the first rope_dim dims: cat(-x2, x1)*sin + x*cos with the
    rotate-half (:137-160) or interleaved (:161-182, sin/cos = first half duplicated pairwise :75-99) layout, remaining
    dims pass through (:184-197); V is copied."""
    B = qkv.shape[0]
    rope_dim = sin.shape[-1]
    half = rope_dim // 2
    q, k, v = qkv.split([q_hidden, kv_hidden, kv_hidden], dim=-1)
    s = sin.reshape(B, 1, rope_dim).float()
    c = cos.reshape(B, 1, rope_dim).float()

    def one(x, w, b):
        x = x.reshape(B, -1, head_dim).float()
        if eps is not None:
            rstd = 1.0 / torch.sqrt((x * x).mean(dim=-1, keepdim=True) + eps)
            x = (x * rstd) * w.float()
            if b is not None:
                x = x + b.float()
        rot, rest = x[..., :rope_dim], x[..., rope_dim:]
        if is_neox_style:
            x1, x2 = rot[..., :half], rot[..., half:]
            cat = torch.cat([-x2, x1], dim=-1)
            out = cat * s + rot * c
        else:
            x1, x2 = rot[..., 0::2], rot[..., 1::2]
            sh, ch = s[..., :half], c[..., :half]
            out = torch.empty_like(rot)
            out[..., 0::2] = (-x2) * sh + x1 * ch
            out[..., 1::2] = x1 * sh + x2 * ch
        return torch.cat([out, rest], dim=-1).reshape(B, -1).to(qkv.dtype)

    return one(q, q_weight, q_bias), one(k, k_weight, k_bias), v.clone()


def split_qkvgate_gemma_rmsnorm_rope(x, sin, cos, q_hidden, kv_hidden, head_dim, rope_dim, eps, q_weight, k_weight):
    """Restates split_qkvgate_gemma_rmsnorm_rope_kernel (python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_rope.py:441-745) in
    fp32: the first 2*q_hidden columns are per-head pairs [q head | gate head] (:478-497); q and k heads: x * rsqrt(mean(x^2) + eps)
    * (w + 1) (:468, :499-503, :586, :603-609), then cat(-x2, x1) * sin + x * cos on the first rope_dim dims, the rest passed through
    (:505-558, :611-656); gate and V are copied (:567-571, :658-667).  PARITY UNPINNED: the reference holds no test or vector for this
    function; tests/test_oracle_kernels.py ties it to split_qkv_rmsnorm_rope (pinned to the reference test's golden) on rearranged input."""
    B = x.shape[0]
    half = rope_dim // 2
    qh = q_hidden // head_dim
    qg = x[:, :2 * q_hidden].reshape(B, qh, 2, head_dim)
    q, gate = qg[:, :, 0, :], qg[:, :, 1, :]
    k = x[:, 2 * q_hidden:2 * q_hidden + kv_hidden].reshape(B, -1, head_dim)
    v = x[:, 2 * q_hidden + kv_hidden:]
    s = sin.reshape(B, 1, rope_dim).float()
    c = cos.reshape(B, 1, rope_dim).float()

    def one(t, w):
        t = t.float()
        rstd = torch.rsqrt((t * t).sum(dim=-1, keepdim=True) / head_dim + eps)
        t = (t * rstd) * (w.float() + 1.0)
        rot, rest = t[..., :rope_dim], t[..., rope_dim:]
        cat = torch.cat([-rot[..., half:], rot[..., :half]], dim=-1)
        return torch.cat([cat * s + rot * c, rest], dim=-1).reshape(B, -1).to(x.dtype)

    return one(q, q_weight), one(k, k_weight), v.clone(), gate.reshape(B, -1).clone()


# --------------------------------------------------------------------------------------
# row statistics / scalings of norm/{l1_norm,rmsnorm_without_weight,rmsnorm_split}.py: transcriptions of the reference tests' goldens
# --------------------------------------------------------------------------------------
def l1_norm(x):
    """tests/python/sgl_kernel_npu/test_l1_norm.py:11-13: fp32 x / sum(x, -1) (kernel: norm/l1_norm.py:19-26)."""
    xf = x.float()
    return xf / xf.sum(dim=-1, keepdim=True)


def rmsnorm_without_weight(x, eps):
    """tests/python/sgl_kernel_npu/test_rmsnorm_without_weight.py:7-11: F.rms_norm without a weight = x * rsqrt(mean(x^2) + eps)
    (kernel: norm/rmsnorm_without_weight.py:50-57), evaluated in fp32 and returned in x's dtype."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)).to(x.dtype)


def fused_variance(x):
    """tests/python/sgl_kernel_npu/test_rmsnorm_split.py:15-16: x.pow(2).mean(-1, keepdim=True) (kernel: norm/rmsnorm_split.py:148-156)."""
    return x.float().pow(2).mean(dim=-1, keepdim=True).to(x.dtype)


def fused_rsqrt_mul(x, variance, weight, eps=1e-6):
    """tests/python/sgl_kernel_npu/test_rmsnorm_split.py:6-12: x * rsqrt(variance + eps) * weight (kernel: norm/rmsnorm_split.py:64-74)."""
    B, L, C = x.shape
    return ((x.float() * torch.rsqrt(variance.float().reshape(B, L, 1) + eps)) * weight.float()).to(x.dtype)


def split_qkv_rmsnorm_mrope(qkv, q_weight, k_weight, cos_sin, num_q_heads, num_kv_heads, head_size, eps, mrope_section, is_interleaved,
                            rope_dim=None, q_bias=None, k_bias=None, has_gate=False):
    """Transcription of the reference test's golden (tests/python/sgl_kernel_npu/test_split_qkv_rmsnorm_mrope.py:7-110: _select_mrope_cos_sin,
    _rms_norm, _apply_mrope, _golden) for norm/split_qkv_rmsnorm_mrope.py:57-420; outputs in the I/O dtype."""
    rope_dim = head_size if rope_dim is None else rope_dim
    T = qkv.shape[0]
    q_size, kv_size = num_q_heads * head_size, num_kv_heads * head_size
    half = rope_dim // 2
    off = torch.arange(half)
    if is_interleaved:
        h_mask = (off % 3 == 1) & (off <= 3 * mrope_section[1])
        w_mask = (off % 3 == 2) & (off <= 3 * mrope_section[2])
        t_mask = ~(h_mask | w_mask)                     # interleaved: t is the complement, nothing is left over
    else:
        t_end = mrope_section[0]
        h_end = t_end + mrope_section[1]
        t_mask = off < t_end
        h_mask = (off >= t_end) & (off < h_end)
        w_mask = (off >= h_end) & (off < h_end + mrope_section[2])
    cs = cos_sin.float()
    # The test's golden sends everything that is neither t nor h to w; the KERNEL masks w as well (contiguous sections: w covers
    # [t + h, t + h + w) only, split_qkv_rmsnorm_mrope.py:157-163, masked loads with other = 0), so rotation offsets behind the three
    # sections get cos = sin = 0.  The two agree whenever the sections fill rope_dim / 2 (every case of the reference test); where they
    # do not, this follows the kernel.
    zero = torch.zeros_like(cs[2, :, :half])
    cos = torch.where(t_mask, cs[0, :, :half], torch.where(h_mask, cs[1, :, :half], torch.where(w_mask, cs[2, :, :half], zero)))
    sin = torch.where(t_mask, cs[0, :, half:], torch.where(h_mask, cs[1, :, half:], torch.where(w_mask, cs[2, :, half:], zero)))
    cos, sin = torch.cat((cos, cos), dim=-1), torch.cat((sin, sin), dim=-1)
    if has_gate:
        q_gate, k, v = qkv.split((q_size * 2, kv_size, kv_size), dim=-1)
        q, gate = q_gate.reshape(T, num_q_heads, head_size * 2).chunk(2, dim=-1)
        gate = gate.reshape(T, q_size).clone()
    else:
        q, k, v = qkv.split((q_size, kv_size, kv_size), dim=-1)
        q = q.reshape(T, num_q_heads, head_size)
        gate = qkv.new_empty((T, 0))
    k = k.reshape(T, num_kv_heads, head_size)

    def norm(x, w, b):
        x = x.float()
        x = x * torch.rsqrt(x.square().mean(dim=-1, keepdim=True) + eps)
        x = x * w.float()
        return x + b.float() if b is not None else x

    def rope(x):
        rot = x[..., :rope_dim]
        x1, x2 = rot.chunk(2, dim=-1)
        rot = rot * cos[:, None, :] + torch.cat((-x2, x1), dim=-1) * sin[:, None, :]
        return torch.cat((rot, x[..., rope_dim:]), dim=-1)

    return (rope(norm(q, q_weight, q_bias)).flatten(1).to(qkv.dtype), rope(norm(k, k_weight, k_bias)).flatten(1).to(qkv.dtype), v.clone(), gate)


def split_qkv_rmsnorm_rope_pos_cache_half(qkv, positions, cos_sin_cache, q_hidden, kv_hidden, head_dim, eps=None, q_weight=None, k_weight=None,
                                          q_bias=None, k_bias=None, rope_dim=None, cast_norm=True):
    """Restates split_qkv_rmsnorm_rope_half_pos_cache_kernel (norm/split_qkv_rmsnorm_rope_pos_cache_half_npu.py:25-230): position clamped to
    the cache (:76-79), cos / sin = the two halves of the cache row in fp32 (:80-91), fp32 x * rsqrt(mean + eps) * w (+ b) (:103-112), optional
    rounding to the I/O dtype (:118-121), o1 = x1 cos - x2 sin, o2 = x2 cos + x1 sin in fp32 (:138-139), rounded on store; V copied.  Pinned:
    tests/test_oracle_kernels.py holds it within the reference test's own tolerance of that test's torch golden (atol 5e-2, rtol 5e-3)."""
    rope_dim = head_dim if rope_dim is None else rope_dim
    B = qkv.shape[0]
    half = rope_dim // 2
    q, k, v = qkv.split([q_hidden, kv_hidden, kv_hidden], dim=-1)
    p = positions.reshape(-1).long().clamp(0, cos_sin_cache.shape[0] - 1)
    cs = cos_sin_cache.float()[p]
    c, s_ = cs[:, None, :half], cs[:, None, half:rope_dim]

    def one(x, w, b):
        y = x.reshape(B, -1, head_dim).float()
        if eps is not None:
            y = y * torch.rsqrt((y * y).sum(dim=-1, keepdim=True) / (1.0 * head_dim) + eps)
            y = y * w.float()[:head_dim] + b.float()[:head_dim] if b is not None else y * w.float()[:head_dim]
        if cast_norm:
            y = y.to(x.dtype).float()
        x1, x2 = y[..., :half], y[..., half:rope_dim]
        out = torch.cat([x1 * c - x2 * s_, x2 * c + x1 * s_, y[..., rope_dim:]], dim=-1)
        return out.reshape(B, -1).to(x.dtype)

    return one(q, q_weight, q_bias), one(k, k_weight, k_bias), v.clone()


def split_qkv_tp_rmsnorm_rope(qkv, cos, sin, q_hidden, kv_hidden, head_dim, eps, q_weight, k_weight, rotary_dim, tp_world=1, other_var=None):
    """Transcription of the reference test's golden (tests/python/sgl_kernel_npu/test_split_qkv_tp_rmsnorm_rope.py:7-44: rms_norm_tp over
    the whole row, rounded to the I/O dtype, then custom_rope with the first half of cos / sin, rounded again), generalised the way the
    kernel is (norm/split_qkv_tp_rmsnorm_rope.py:75-177): rotary_dim <= head_dim, and `other_var` [B, 2] = the sum of the OTHER ranks' local
    means of squares (what the all-reduce adds) for tp_world > 1."""
    B = qkv.shape[0]
    q, k, v = qkv.split([q_hidden, kv_hidden, kv_hidden], dim=-1)
    half = rotary_dim // 2
    c = cos.reshape(B, 1, rotary_dim).float()[..., :half]
    s_ = sin.reshape(B, 1, rotary_dim).float()[..., :half]

    def one(x, w, col):
        xf = x.float()
        var = xf.pow(2).mean(dim=-1, keepdim=True)
        if other_var is not None:
            var = var + other_var[:, col:col + 1].float()
        y = (xf * (1.0 / torch.sqrt(var * (1.0 / tp_world) + eps)) * w.float()).to(x.dtype).float().reshape(B, -1, head_dim)
        x1, x2 = y[..., :half], y[..., half:2 * half]
        out = torch.cat([x1 * c - x2 * s_, x2 * c + x1 * s_, y[..., 2 * half:]], dim=-1)
        return out.reshape(B, -1).to(x.dtype)

    return one(q, q_weight, 0), one(k, k_weight, 1), v.clone()


def fused_split_qk_norm(x, q_weight, q_bias, k_weight, k_bias, q_lora_rank, kv_lora_rank, qk_rope_dim, eps=1e-6):
    """Restates fused_split_qk_norm_kernel (norm/fused_split_qk_norm.py:6-91) in fp32: per part x * rsqrt(sum(x^2) / n + eps) * w (+ b)
    (:39-47, :62-70), the rope part copied (:75-90).  PARITY UNPINNED (the reference holds no test for it); tests/test_oracle_kernels.py ties
    it to the pinned add_rmsnorm_bias oracle."""
    q, kn, kp = x.split([q_lora_rank, kv_lora_rank, qk_rope_dim], dim=-1)

    def one(t, w, b):
        tf = t.float()
        y = (tf * torch.rsqrt((tf * tf).sum(dim=-1, keepdim=True) / t.shape[-1] + eps)) * w.float()
        return (y + b.float() if b is not None else y).to(x.dtype)

    return one(q, q_weight, q_bias), one(kn, k_weight, k_bias).unsqueeze(1), kp.clone().unsqueeze(1)


def attention_sinks(query, k_cache, v_cache, sinks, block_tables, kv_lens, scale, window, q_head_num, k_head_num, bt_rows=None):
    """Restates attention_sinks_kernel / attention_sinks_prefill_kernel (attention/sinks_attention.py:7-87, :139-238) in fp32, one query row at
    a time: keys [max(len - window, 0), len) of the row's sequence (:35-39, :174-180), logits q . k * scale, softmax over the keys AND the
    head's sink logit (running maximum starts at the sink, :45, the denominator gains exp(sink - max), :78-79), P rounded to the cache dtype
    before P . V (:73).  kv_lens [rows]; bt_rows [rows] = block-table row of a query row (extend form).  PARITY UNPINNED (no reference test);
    tests/test_oracle_kernels.py ties it to the pinned decode_gqa oracle (sink = -inf, no window)."""
    rows = query.shape[0]
    D = query.shape[1] // q_head_num
    page = k_cache.shape[1]
    Dv = v_cache.shape[-1]
    group = q_head_num // k_head_num
    out = torch.zeros(rows, q_head_num, Dv, dtype=torch.float32)
    q = query.reshape(rows, q_head_num, D).float()
    for r in range(rows):
        n = int(kv_lens[r])
        lo = max(n - window, 0) if window != -1 else 0
        br = int(bt_rows[r]) if bt_rows is not None else r
        pos = torch.arange(lo, n)
        blk = block_tables[br][pos // page].long()
        for h in range(q_head_num):
            kvh = h // group
            sk = float(sinks[h])
            if n > lo:
                k = k_cache[blk, pos % page, kvh].float()
                v = v_cache[blk, pos % page, kvh]
                s = (k @ q[r, h]) * scale
                m = max(float(s.max()), sk)
                pexp = torch.exp(s - m)
                l = float(pexp.sum()) + float(torch.exp(torch.tensor(sk - m)))
                out[r, h] = (pexp.to(v.dtype).float() @ v.float()) / l
    return out.to(query.dtype).reshape(rows, q_head_num * Dv)


def fia_prep(topk_idx, seq_lens, per_query_req, req_to_token, block_size):
    """Restates _fia_prep_kernel (attention/fia_blockq_attention.py:35-88) query by query: the own logical block (position // block_size) goes
    last, the other real blocks keep their order, logical -> physical page through req_to_token[req, block * block_size] // block_size (:58-72),
    pads 0; actual_kvlen = real non-own blocks * block_size (+ offset + 1 when the own block is selected) (:84-87).  PARITY UNPINNED (no
    reference test).  -> (block_table [T, topk1] int32, actual_kvlen [T] int32)."""
    T, topk1 = topk_idx.shape
    max_cols = req_to_token.shape[1]
    bt = torch.zeros((T, topk1), dtype=torch.int32)
    kvl = torch.zeros(T, dtype=torch.int32)
    for t in range(T):
        abs_pos = max(int(seq_lens[t]) - 1, 0)
        own, own_offset = abs_pos // block_size, abs_pos % block_size
        req = int(per_query_req[t])
        blocks = [int(b) for b in topk_idx[t]]
        real = [b for b in blocks if b >= 0 and b != own]
        own_present = any(b == own for b in blocks if b >= 0)
        page = lambda b: int(req_to_token[req, min(b * block_size, max_cols - 1)]) // block_size
        for i, b in enumerate(real):
            bt[t, i] = page(b)
        if len(real) < topk1:
            bt[t, len(real)] = page(own)
        kvl[t] = len(real) * block_size + (own_offset + 1 if own_present else 0)
    return bt, kvl


def fia_blockq_sparse(q, k_cache, v_cache, topk_idx, seq_lens, per_query_req, req_to_token, block_size, sm_scale):
    """fp32 softmax attention of every query over the first actual_kvlen keys of its own block table (what the reference asks of its
    fused-infer-attention op, fia_blockq_attention.py:167-180)."""
    bt, kvl = fia_prep(topk_idx, seq_lens, per_query_req, req_to_token, block_size)
    T, Hq, D = q.shape
    out = torch.zeros((T, Hq, v_cache.shape[-1]), dtype=torch.float32)
    for t in range(T):
        n = int(kvl[t])
        if n == 0:
            continue
        pages = bt[t, :(n + block_size - 1) // block_size].long()
        k = k_cache[pages, :, 0, :].reshape(-1, D)[:n].float()
        v = v_cache[pages, :, 0, :].reshape(-1, v_cache.shape[-1])[:n].float()
        p = torch.softmax((q[t].float() @ k.T) * sm_scale, dim=-1)
        out[t] = p @ v
    return out.to(q.dtype)


def swiglu_oai(x, dim, alpha, limit):
    """Restates swiglu_oai_kernel (activation/swiglu_oai.py:7-50) in fp32: gate = even columns clamped from above (:36), up = odd columns
    clamped to +-limit (:37-38), (up + 1) * gate * 1 / (1 + exp(-gate * alpha)) (:39-41).  Pinned to the file's own torch formulation
    swiglu_oai_native (:86-96), which computes in the tensors' dtype (tests/test_oracle_kernels.py: exact for fp32 inputs)."""
    xf = x.reshape(-1, dim).float()
    gate, up = xf[:, 0::2].clamp(max=limit), xf[:, 1::2].clamp(min=-limit, max=limit)
    return ((up + 1.0) * (gate * (1.0 / (1.0 + torch.exp(-gate * alpha))))).to(x.dtype)


def swiglu_oai_quant(x, alpha, limit, need_quant=True, total_rows=None):
    """Restates _swiglu_oai_quant_kernel (activation/swiglu_oai_quant.py:39-112) in fp32: gate = min(x1, limit), up = clamp(x2, +-limit),
    out = gate * sigmoid(gate * alpha) * (up + 1) (:83-85); scale = max|out| / 127 (:89); q = saturate(trunc(dtype(out / scale))) (:96-97 --
    the cast's rounding is the backend's in the reference; truncation = the Triton language's float -> int conversion: stated assumption,
    PARITY UNPINNED, no reference test exists).  Rows >= total_rows are returned as zeros (uninitialised in the kernel)."""
    x2 = x.reshape(-1, x.shape[-1]).float()
    half = x2.shape[-1] // 2
    gate, up = x2[:, :half].clamp(max=limit), x2[:, half:].clamp(min=-limit, max=limit)
    out = (gate * (1.0 / (1.0 + torch.exp(-gate * alpha)))) * (up + 1.0)
    n = x2.shape[0] if total_rows is None else total_rows
    if not need_quant:
        o = out.to(x.dtype)
        o[n:] = 0
        return o.reshape(*x.shape[:-1], half), None
    scale = out.abs().amax(dim=-1) / 127.0
    q = (out / scale[:, None]).to(x.dtype).float()
    q = torch.nan_to_num(q, nan=0.0).trunc().clamp(-128, 127).to(torch.int8)
    q[n:] = 0
    scale = scale.clone()
    scale[n:] = 0
    return q.reshape(*x.shape[:-1], half), scale


def situ_and_mul(x, beta=4.0, linear_beta=25.0, need_quant=False, total_rows=None):
    """Restates the SiTU kernels (activation/situ.py:58-90, :101-162, :395-427) in fp32: gate' = beta * tanh(gate / beta) * sigmoid(gate) (:61),
    up' = linear_beta * tanh(up / linear_beta) when linear_beta is given (:62-63), out = gate' * up' (:64); quantised: scale = max(max|out| / 127,
    1e-30) (:67), q = clamp(floor(out / scale + 0.5), -128, 127) (:78-80).  PARITY UNPINNED: the reference holds no test or vector for this file.
    Rows >= total_rows are returned as zeros (not written by the kernel)."""
    x2 = x.reshape(-1, x.shape[-1]).float()
    half = x2.shape[-1] // 2
    gate, up = x2[:, :half], x2[:, half:]
    ga = (beta * torch.tanh(gate * (1.0 / beta))) * (1.0 / (1.0 + torch.exp(-gate)))
    if linear_beta is not None:
        up = linear_beta * torch.tanh(up * (1.0 / linear_beta))
    out = ga * up
    n = x2.shape[0] if total_rows is None else total_rows
    if not need_quant:
        o = out.to(x.dtype)
        o[n:] = 0
        return o.reshape(*x.shape[:-1], half), None
    scale = torch.clamp(out.abs().amax(dim=-1) / 127.0, min=1e-30)
    q = torch.floor(out / scale[:, None] + 0.5).clamp(-128, 127).to(torch.int8)
    q[n:] = 0
    scale = scale.clone()
    scale[n:] = 0
    return q.reshape(*x.shape[:-1], half), scale


def attn_residual_mix(prefix_sum, bank, num_valid_blocks, combined_weight, eps):
    """Restates _mix_fused_kernel (kimi_k3/attn_residual.py:27-63) in fp32: rows = bank[:, :B] then the prefix row; score = sum(row *
    rsqrt(mean(row^2) + eps) * combined_weight) (:40-41), softmax over the B + 1 scores (:43-45), out = sum_r p_r row_r (:47-59).  PARITY
    UNPINNED (no reference test)."""
    rows = torch.cat([bank[:, :num_valid_blocks].float(), prefix_sum[:, None].float()], dim=1)          # [T, B + 1, H]
    inv = torch.rsqrt((rows * rows).sum(dim=-1) / rows.shape[-1] + eps)
    scores = ((rows * inv[..., None]) * combined_weight.float()).sum(dim=-1)
    p = torch.softmax(scores, dim=-1)
    return (p[..., None] * rows).sum(dim=1).to(prefix_sum.dtype)


def mul_add(routed, shared, factor):
    """moe/mul_add.py:24-26: routed * factor + shared evaluated in the tensors' dtype (the product is rounded before the sum).  PARITY
    UNPINNED (no reference test)."""
    return (routed.float() * factor).to(routed.dtype).float().add(shared.float()).to(routed.dtype)


def zero_experts_compute_identity(expert_indices, expert_scales, num_experts, hidden, identity_mask_value=0):
    """Restates experts_compute_identity_kernel (moe/zero_experts_compute_identity.py:20-47): -> (result, indices, scales) with the in-place
    effects applied to copies.  PARITY UNPINNED (no reference test)."""
    idx, sc = expert_indices.clone(), expert_scales.clone()
    mask = idx >= num_experts                                           # :24
    sum_scales = torch.where(mask, sc.float(), torch.zeros_like(sc, dtype=torch.float32)).sum(dim=-1)      # :25-26
    result = (hidden.float() * sum_scales[:, None]).to(hidden.dtype)    # :35-39
    ident = torch.full_like(idx, identity_mask_value)
    all_zero = mask.all(dim=-1)                                         # :29-31: no real expert left: the first selection becomes expert 0
    ident[all_zero, 0] = 0
    sc[mask] = 0                                                        # :40
    idx[mask] = ident[mask]                                             # :41
    return result, idx, sc


def fused_scale_shift(x, scale, shift, scale_constant=1.0):
    """tests/python/sgl_kernel_npu/test_scale_shift.py:6-11: x * (1 + scale) + shift; with one shift value per element the kernel uses
    scale_constant instead of 1 (norm/scale_shift.py:112 against :60).  fp32, returned in x's dtype."""
    full = shift.numel() == x.numel()                    # tested first by the wrapper (:149)
    c = scale_constant if full else 1.0
    sc = scale.float().reshape(-1)
    sh = shift.float().reshape(-1)
    sh = sh.reshape(x.shape) if full else sh
    return (x.float() * (c + sc) + sh).to(x.dtype)


# --------------------------------------------------------------------------------------
# A14  mla_preprocess
# --------------------------------------------------------------------------------------
def fused_rope_qk_mqa(query, key, cos_sin, rotary_dim, is_neox_style):
    """Transcription of the reference test golden forward_native / apply_rotary_emb
    (tests/python/sgl_kernel_npu/test_fused_rope_qk_mqa.py:5-60): torch ops in the I/O dtype (every product and sum rounded),
    cos = cos_sin[:, :R/2], sin = cos_sin[:, R/2:]; the first rotary_dim dims rotated, the rest passed through."""
    cos, sin = cos_sin[:, :rotary_dim].chunk(2, dim=-1)

    def rot(x):
        xr, xp = x[..., :rotary_dim], x[..., rotary_dim:]
        c, s_ = cos.unsqueeze(-2).to(x.dtype), sin.unsqueeze(-2).to(x.dtype)
        if is_neox_style:
            x1, x2 = torch.chunk(xr, 2, dim=-1)
        else:
            x1, x2 = xr[..., ::2], xr[..., 1::2]
        o1, o2 = x1 * c - x2 * s_, x2 * c + x1 * s_
        o = torch.cat((o1, o2), dim=-1) if is_neox_style else torch.stack((o1, o2), dim=-1).flatten(-2)
        return torch.cat((o, xp), dim=-1)

    return rot(query), rot(key)


def _rotate_half(x):
    a, b = torch.chunk(x, 2, dim=-1)
    return torch.cat([-b, a], dim=-1)


def _quant_per_tensor(x, scale, zp):
    """tests/python/sgl_kernel_npu/test_mla_preprocess.py:77-83."""
    x = x / scale.float() + zp.float()
    x = torch.clamp(x.to(torch.float16), -128, 127)
    return torch.round(x).to(torch.int8)


def _quant_per_tensor_muls(x, scale, zp):
    """tests/python/sgl_kernel_npu/test_mla_preprocess.py:83-90 (multiply instead of divide: the q_nope quantisation of int8_nzcache)."""
    x = x * scale.float() + zp.float()
    x = torch.clamp(x.to(torch.float16), -128, 127)
    return torch.round(x).to(torch.int8)


def nz_cache_offsets(slot, block_size, dim, c0):
    """Flat element offsets (into a cache [blocks, block_size, 1, dim]) of the `dim` values of cache slot `slot` in the NZ layouts:
    inside the slot's block, element d sits at ((d // c0) * block_size + slot % block_size) * c0 + d % c0 -- the positions the
    reference test reads back (extract_from_nzcache, tests/python/sgl_kernel_npu/test_mla_preprocess.py:122-136, which hard-codes
    block_size = 128; c0 = 16 for bf16 / fp16, 32 for int8)."""
    blk, inner = slot // block_size, slot % block_size
    d = torch.arange(dim)
    return blk * block_size * dim + ((d // c0) * block_size + inner) * c0 + d % c0


def extract_from_nzcache(cache, slot, c0):
    """Read one slot's row back out of an NZ cache [blocks, block_size, 1, dim] (the reference test's read-back, :122-136)."""
    block_size, dim = cache.shape[1], cache.shape[-1]
    return cache.reshape(-1)[nz_cache_offsets(int(slot), block_size, dim, c0)]


def _int8_gemm_dequant(a, w, descale, bias, dtype):
    """:95-107 (bf16 branch): exact int32 GEMM + bias, * float descale, to dtype."""
    # exact: |sum| <= K * 128 * 128 < 2^53, so the float64 BLAS product is the integer product (torch's int32 matmul is a scalar loop)
    y = torch.round(a.double() @ w.double().t()).to(torch.int32)
    if bias is not None and bias.numel():
        y = y + bias
    return (y.to(torch.float32) * descale.float()).to(dtype)


def _quant_per_token(x):
    """quant_mode 'per_token_quant_symm' (csrc/mla_preprocess/op_kernel/mla_preprocess_mix_bf16.hpp:437-483): scale = max|x| / 127 per
    row (fp32), y = x * (1 / scale), fp16, clamp, round half to even, int8.  -> (int8, float32 scales).  An all-zero row gives zeros
    with scale 0 (the reference would divide by zero)."""
    xf = x.float()
    scale = xf.abs().amax(dim=-1) / torch.full((), 127.0)
    inv = torch.where(scale > 0, torch.ones_like(scale) / scale, torch.zeros_like(scale))
    y = torch.clamp((xf * inv[..., None]).to(torch.float16), -128, 127)
    return torch.round(y).to(torch.int8), scale


def _int8_gemm_dequant_token(a, w, descale, tok_scale, dtype):
    """:389-421 / :2262-2279: (float(int32) * per-channel descale) * per-token scale, no bias, one rounding to dtype."""
    y = torch.round(a.double() @ w.double().t()).to(torch.int32)
    return ((y.to(torch.float32) * descale.float()) * tok_scale.float()[:, None]).to(dtype)


def mla_preprocess_per_token(hidden, wdqkv, descale0, gamma1, beta1, gamma2, wuq, descale1, wuk, cos, sin, eps=1e-6):
    """mla_preprocess with quant_mode='per_token_quant_symm' (the reference's default; no golden exists for it in the reference
    tests -- tests/python/sgl_kernel_npu/test_mla_preprocess.py only runs per_tensor_quant_asymm -- so this restates the kernel:
    mla_preprocess_mix_bf16.hpp Quant :284-490, the GEMM2 epilogue :2196-2290; PARITY UNPINNED for this mode).  Same network as
    mla_preprocess() below with the two quantisations per token and the two dequants by (channel scale) * (token scale), no bias."""
    dtype = hidden.dtype
    N = hidden.shape[0]
    Hq = wuq.shape[0] // 192

    def rms(x, g):
        xf = x.float()
        return xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * g.float()

    a8, t0 = _quant_per_token(hidden)
    fused = _int8_gemm_dequant_token(a8, wdqkv, descale0, t0, dtype)
    latent, q = fused.split([576, 1536], dim=-1)
    k_nope, k_pe = latent[..., :512], latent[..., 512:].unsqueeze(1)
    q = rms(q, gamma1) + beta1
    k_nope = rms(k_nope, gamma2)
    q8, t1 = _quant_per_token(q)
    q_out = _int8_gemm_dequant_token(q8, wuq, descale1, t1, dtype).view(N, Hq, 192)
    q_nope, q_pe = q_out.split([128, 64], dim=-1)
    q_nope_out = torch.bmm(q_nope.transpose(0, 1), wuk).transpose(0, 1)
    c, s = cos.unsqueeze(1).float(), sin.unsqueeze(1).float()
    q_pe_r = (q_pe.float() * c + _rotate_half(q_pe.float()) * s).to(dtype)
    k_pe_r = (k_pe.float() * c + _rotate_half(k_pe.float()) * s).to(dtype)
    return q_nope_out.to(dtype), q_pe_r, k_nope.to(dtype), k_pe_r.squeeze(1)


def mla_preprocess(hidden, wdqkv, descale0, bias0, gamma1, beta1, gamma2, wuq, descale1, bias1, wuk, cos, sin, qscale0, qoff0,
                   qscale1, qoff1, eps=1e-6, cache_mode="krope_ctkv", ctkv_scale=None, qnope_scale=None):
    """Transcription of golden2_pytorch (tests/python/sgl_kernel_npu/test_mla_preprocess.py:407-483).  cache_mode 'krope_ctkv' and
    'nzcache' produce the same VALUES (only the cache layout differs, see nz_cache_offsets); 'int8_nzcache' (:465-475) returns
    q_nope_out and k_nope as int8: quant_per_tensor_muls(q_nope_out, qnope_scale[head]) and quant_per_tensor(k_nope, ctkv_scale).
    The network:
    quant -> INT8 GEMM [N,H]x[H,2112] -> split [512 k_nope | 64 k_pe | 1536 q] -> rms_norm(q)*gamma1+beta1,
    rms_norm(k_nope)*gamma2 -> quant -> INT8 GEMM [N,1536]x[1536,Hq*192] -> per head [128|64] -> bmm(q_nope, wuk) ->
    rotate-half RoPE on q_pe / k_pe.  Weights are [out, in] row-major.  Returns (q_nope_out [N,Hq,512], q_pe [N,Hq,64],
    k_nope [N,512], k_pe [N,64]) in hidden.dtype (what the op writes to q_out0, q_out1 and the two caches)."""
    dtype = hidden.dtype
    N = hidden.shape[0]
    Hq = wuq.shape[0] // 192

    def rms(x, g):
        xf = x.float()
        return xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * g.float()

    fused = _int8_gemm_dequant(_quant_per_tensor(hidden, qscale0, qoff0), wdqkv, descale0, bias0, dtype)
    latent, q = fused.split([576, 1536], dim=-1)
    k_nope, k_pe = latent[..., :512], latent[..., 512:].unsqueeze(1)
    q = rms(q, gamma1) + beta1
    k_nope = rms(k_nope, gamma2)
    q_out = _int8_gemm_dequant(_quant_per_tensor(q, qscale1, qoff1), wuq, descale1, bias1, dtype).view(N, Hq, 192)
    q_nope, q_pe = q_out.split([128, 64], dim=-1)
    q_nope_out = torch.bmm(q_nope.transpose(0, 1), wuk).transpose(0, 1)
    c, s = cos.unsqueeze(1).float(), sin.unsqueeze(1).float()
    q_pe_r = (q_pe.float() * c + _rotate_half(q_pe.float()) * s).to(dtype)
    k_pe_r = (k_pe.float() * c + _rotate_half(k_pe.float()) * s).to(dtype)
    if cache_mode == "int8_nzcache":
        q8 = _quant_per_tensor_muls(q_nope_out, qnope_scale.reshape(1, Hq, 1), torch.zeros_like(q_nope_out))
        k8 = _quant_per_tensor(k_nope, ctkv_scale, torch.zeros_like(k_nope))
        return q8, q_pe_r, k8, k_pe_r.squeeze(1)
    return q_nope_out.to(dtype), q_pe_r, k_nope.to(dtype), k_pe_r.squeeze(1)
