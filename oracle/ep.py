"""NumPy restatement of the DeepEP dispatch/combine arithmetic of sgl-kernel-npu.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All ranks of the EP group are
simulated inside one process: functions take per-rank lists and return per-rank
lists.  Integer/index results are the bit-exact target for the HIP kernels; the
INT8 payload / scales and the BF16 combine are bit-exact targets too (fp32
arithmetic with the rounding modes of the reference, stated per function).

Notation: W ranks, E experts, L = E // W local experts,
T tokens (per rank), K = top-k, H = hidden.  "idx i" over [L*W] means
i = local_expert * W + src_rank (reference: cam_moe_dispatch_normal.h:723-724).

bf16 tensors are carried as uint16 bit patterns (oracle/bf16.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from .bf16 import bf16_bits_to_f32, f32_to_bf16_bits_rne


# --------------------------------------------------------------------------------------
# A1  dispatch layout
# --------------------------------------------------------------------------------------
def dispatch_layout(topk_idx: np.ndarray, num_experts: int, num_ranks: int) -> dict:
    """Reference: csrc/deepep/ops/op_kernel/dispatch_layout.h:159-213 (+ host
    csrc/deepep/deep_ep.cpp:111-180).

    For every token row: ids < 0 or >= E are skipped (:166); num_tokens_per_expert[e]++
    (:169-170); the first hit of rank e // L sets is_token_in_rank[t, r] = 1 and bumps
    num_tokens_per_rank[r] (:171-177).  Second pass (:203-213): send_token_idx_small[t, k] =
    number of earlier (row-major) valid (t', k') pairs with the same expert.  Entries of
    send_token_idx_small at invalid ids are unspecified in the reference (UB scratch written
    back); this restatement (and the HIP kernel) define them as 0.
    """
    topk_idx = np.ascontiguousarray(topk_idx, dtype=np.int64)
    T, K = topk_idx.shape
    E, W = int(num_experts), int(num_ranks)
    L = E // W
    flat = topk_idx.reshape(-1)
    valid = (flat >= 0) & (flat < E)
    vidx = np.nonzero(valid)[0]
    ve = flat[vidx]
    num_tokens_per_expert = np.bincount(ve, minlength=E).astype(np.int32)
    # rank among same-expert pairs in row-major order == position inside the stable sort group
    order = np.argsort(ve, kind="stable")
    starts = np.zeros(E + 1, dtype=np.int64)
    np.cumsum(num_tokens_per_expert, out=starts[1:])
    pos_sorted = np.arange(ve.size, dtype=np.int64) - starts[ve[order]]
    small = np.zeros(T * K, dtype=np.int32)
    small[vidx[order]] = pos_sorted.astype(np.int32)
    # token -> rank membership (dedup over k)
    is_in = np.zeros((T, W), dtype=np.int32)
    if vidx.size:
        is_in[vidx // K, ve // L] = 1
    num_tokens_per_rank = is_in.sum(axis=0).astype(np.int32)
    return dict(
        num_tokens_per_rank=num_tokens_per_rank,
        num_tokens_per_expert=num_tokens_per_expert,
        is_token_in_rank=is_in,
        send_token_idx_small=small.reshape(T, K),
    )


def dispatch_layout_loops(topk_idx: np.ndarray, num_experts: int, num_ranks: int) -> dict:
    """Literal (slow) transcription of the same two passes with Python loops; used to
    cross-check the vectorised version on small cases."""
    T, K = topk_idx.shape
    E, W = num_experts, num_ranks
    L = E // W
    per_e = np.zeros(E, np.int32)
    per_r = np.zeros(W, np.int32)
    is_in = np.zeros((T, W), np.int32)
    for t in range(T):
        seen = [False] * W
        for k in range(K):
            e = int(topk_idx[t, k])
            if e < 0 or e >= E:
                continue
            per_e[e] += 1
            r = e // L
            if not seen[r]:
                seen[r] = True
                is_in[t, r] = 1
                per_r[r] += 1
    run = np.zeros(E, np.int32)
    small = np.zeros((T, K), np.int32)
    for t in range(T):
        for k in range(K):
            e = int(topk_idx[t, k])
            if e < 0 or e >= E:
                continue
            small[t, k] = run[e]
            run[e] += 1
    return dict(num_tokens_per_rank=per_r, num_tokens_per_expert=per_e, is_token_in_rank=is_in,
                send_token_idx_small=small)


# --------------------------------------------------------------------------------------
# A2  notify dispatch (counts exchange + derived index tables)
# --------------------------------------------------------------------------------------
def send_data_offset(num_tokens_per_expert: np.ndarray) -> np.ndarray:
    """Exclusive prefix over global expert id of this rank's per-expert counts
    (reference: notify_dispatch.h:185-198, `sendDataOffsetTensor(i) = prefixSum`)."""
    c = np.asarray(num_tokens_per_expert, dtype=np.int64)
    out = np.zeros_like(c)
    np.cumsum(c[:-1], out=out[1:])
    return out.astype(np.int32)


def notify_dispatch(cnt: np.ndarray, num_tokens: Sequence[int], rank: int) -> dict:
    """Tables rank `rank` derives after the all-to-all of (count, send-prefix, roundTokens)
    triples.  `cnt[src, e]` = num_tokens_per_expert of rank src (all ranks), round = 1.

    Reference: csrc/deepep/ops/op_kernel/notify_dispatch.h
      recv_count  :473-482  inclusive running sum over idx i = le*W+src of cnt[src, me*L+le]
      recv_offset :386-407  the sender's exclusive prefix for that expert
      recv_tokens_per_expert :606-615, expert_global_offset :665-669 (exclusive scan),
      srcrank_in_expert_offset :715-721 (exclusive scan over src inside one le),
      r_in_srcrank_offset :759-780 (0 for round 1), total_recv_token :434-450,
      max_bs :553-577 (max over src of that src's token count).
    """
    cnt = np.asarray(cnt, dtype=np.int64)
    W, E = cnt.shape
    L = E // W
    me = int(rank)
    c = cnt[:, me * L:(me + 1) * L].T.copy()          # [L, W]  c[le, src]
    send_off = np.zeros_like(cnt)
    np.cumsum(cnt[:, :-1], axis=1, out=send_off[:, 1:])
    recv_count = np.cumsum(c.reshape(-1)).astype(np.int32)                       # [L*W]
    recv_offset = send_off[:, me * L:(me + 1) * L].T.reshape(-1).astype(np.int32)  # [L*W]
    per_e = c.sum(axis=1)
    ego = np.zeros(L, np.int64)
    np.cumsum(per_e[:-1], out=ego[1:])
    sie = np.zeros((L, W), np.int64)
    np.cumsum(c[:, :-1], axis=1, out=sie[:, 1:])
    return dict(
        send_data_offset=send_off[me].astype(np.int32),
        recv_count=recv_count,
        recv_offset=recv_offset,
        recv_tokens_per_expert=per_e.astype(np.int32),
        expert_global_offset=ego.astype(np.int32),
        srcrank_in_expert_offset=sie.reshape(-1).astype(np.int32),
        r_in_srcrank_offset=np.zeros(L * W, np.int32),
        total_recv_token=np.int32(c.sum()),
        max_bs=np.int32(max(int(t) for t in num_tokens) if len(num_tokens) else 0),
    )


# --------------------------------------------------------------------------------------
# per-token dynamic INT8 quantisation
# --------------------------------------------------------------------------------------
def quant_int8_rows(x_bits: np.ndarray, eps: Optional[float]) -> tuple:
    """Per-row symmetric INT8 quantisation of bf16 rows.

    Normal mode (eps = 1e-12): cam_moe_dispatch_normal.h:326-363
        s = 127.0f / (max_j |x_j| + 1e-12f);  q_j = int8(rint_half_even(float(x_j) * s));
        scale_out = 1.0f / s
      (bf16->f32 exact; f32 multiply; f32->i32 CAST_RINT = round half to even; the later
       i32->f16->i8 casts are exact for |v| <= 127.)
    Low-latency mode (eps = None): moe_distribute_dispatch_v2.h:1006-1033, same but
        s = 127.0f / max_j |x_j| with no epsilon.  An all-zero row gives s = inf and
        0 * inf = NaN in the reference (result unspecified); here (and in the HIP kernel)
        such a row quantises to q = 0, scale = 0.
    Non-finite input (the reference's ReduceMax over a row holding a NaN is unspecified): a row that holds a
    NaN or an infinity has max |x| = +inf here and in the HIP kernel (which takes the maximum on the bf16 bit
    patterns, where a NaN orders above infinity and is clamped to it) -> s = 0, scale_out = +inf, q = 0 at
    every finite element; the bytes at the non-finite positions themselves are unspecified.
    Returns (q int8 [N,H], scale f32 [N]).
    """
    xf = bf16_bits_to_f32(x_bits)
    with np.errstate(invalid="ignore"):
        ax = np.abs(xf)
        ax = np.where(np.isnan(ax), np.float32(np.inf), ax)
    amax = np.max(ax, axis=1).astype(np.float32) if xf.shape[1] else np.zeros(xf.shape[0], np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if eps is None:
            s = np.float32(127.0) / amax
        else:
            s = np.float32(127.0) / (amax + np.float32(eps))
        y = xf * s[:, None]
        q = np.nan_to_num(np.rint(y), nan=0.0)   # round half to even (non-finite positions: unspecified, 0 here)
        scale = (np.float32(1.0) / s).astype(np.float32)
    if eps is None:
        zero = amax == 0
        if zero.any():
            q[zero] = 0
            scale[zero] = 0
    return q.astype(np.int8), scale


# --------------------------------------------------------------------------------------
# per-token FP8 E4M3 quantisation (quant_mode "pertoken_fp8_e4m3")  -- PARITY UNPINNED: the reference serves this mode on its
# Ascend950 build only (deep_ep.cpp:338-343), holds no vector for it, and the kernel below is restated from source.
# --------------------------------------------------------------------------------------
def f32_to_e4m3fn_bits(y: np.ndarray) -> np.ndarray:
    """float32 -> OCP FP8 E4M3 ("fn": no infinities, S.1111.111 = NaN, largest finite 448 = S.1111.110), round to nearest even,
    as a uint8 bit pattern.  Written from the format definition (OCP 8-bit Floating Point Specification v1.0, section 5): bias 7,
    3 mantissa bits, subnormals below 2^-6 in steps of 2^-9.  Magnitudes that round above 448 saturate to 448 here (they cannot
    occur in quant_fp8_e4m3_rows: |x * s| <= 448 up to one fp32 rounding, far below the midpoint 464 to the next binade step)."""
    y = np.asarray(y, np.float32)
    sign = (y.view(np.uint32) >> 31).astype(np.uint8) << 7
    a = np.abs(y).astype(np.float64)
    out = np.zeros(y.shape, np.uint8)
    nan = np.isnan(a)
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))            # exponent of the binade (exact for floats: log2 of a power of two
        e = np.where(np.ldexp(1.0, e.astype(np.int64)) > a, e - 1, e)      # is exact, the correction covers the neighbours)
        e = np.where(np.ldexp(1.0, (e + 1).astype(np.int64)) <= a, e + 1, e)
    e = np.clip(e, -6, 8)                                            # below 2^-6: the subnormal step 2^-9 applies
    step = np.ldexp(1.0, (e - 3).astype(np.int64))                   # spacing of representable values in that binade
    qn = np.rint(a / step)                                           # np.rint: half to even; a / step is exact (power of two)
    val = qn * step
    val = np.minimum(val, 448.0)
    # value -> bits: normal numbers have qn in [8, 16) (16 = next binade), subnormals qn in [0, 8)
    e2 = np.floor(np.log2(np.where(val > 0, val, 1.0)))
    e2 = np.where(np.ldexp(1.0, e2.astype(np.int64)) > val, e2 - 1, e2)
    e2 = np.where(np.ldexp(1.0, (e2 + 1).astype(np.int64)) <= val, e2 + 1, e2)
    normal = val >= 2.0 ** -6
    mant_n = np.rint(val / np.ldexp(1.0, (e2 - 3).astype(np.int64))) - 8
    bits_n = ((e2 + 7).astype(np.int64) << 3) | mant_n.astype(np.int64)
    bits_s = np.rint(val / 2.0 ** -9).astype(np.int64)
    out = np.where(normal, bits_n, bits_s).astype(np.uint8)
    out = np.where(val == 0, 0, out).astype(np.uint8)
    out = np.where(nan, 0x7F, out).astype(np.uint8)
    return out | sign


def e4m3fn_bits_to_f32(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b, np.uint8).astype(np.int64)
    s = np.where(b & 0x80, -1.0, 1.0)
    e, m = (b >> 3) & 0xF, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (8 + m) * np.ldexp(1.0, e - 10))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return (s * v).astype(np.float32)


def quant_fp8_e4m3_rows(x_bits: np.ndarray) -> tuple:
    """Per-row FP8 E4M3 quantisation of bf16 rows (moe_distribute_dispatch_v2_a5.h:1109-1157, QuantDynamicPerToken with
    ExpandXOutType = fp8_e4m3fn): tokenF32 = float(x); scale = max|x| > 0 ? 448.0f / max|x| : 1.0f (:1130-1131); tokenF32 *= scale
    (:1133); out = cast<e4m3fn, CAST_RINT>(tokenF32) (:1150); the row's scale word = 1.0f / scale (:1154-1155).
    Non-finite rows as in quant_int8_rows (max = +inf -> scale 0, scale word +inf).  Returns (uint8 e4m3 bits [N,H], f32 [N])."""
    xf = bf16_bits_to_f32(x_bits)
    with np.errstate(invalid="ignore"):
        ax = np.abs(xf)
        ax = np.where(np.isnan(ax), np.float32(np.inf), ax)
    amax = np.max(ax, axis=1).astype(np.float32) if xf.shape[1] else np.zeros(xf.shape[0], np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        s = np.where(amax > 0, np.float32(448.0) / np.where(amax > 0, amax, np.float32(1)), np.float32(1.0)).astype(np.float32)
        y = (xf * s[:, None]).astype(np.float32)
        scale = (np.float32(1.0) / s).astype(np.float32)
    return f32_to_e4m3fn_bits(np.nan_to_num(y, nan=0.0)), scale


def _quant_rows(x_bits, quant, eps):
    """quant: False / True ("int8") / "fp8" (pertoken_fp8_e4m3) -> (payload [N,H] int8 | uint8 e4m3 bits, scales f32 [N])."""
    if quant == "fp8":
        return quant_fp8_e4m3_rows(x_bits)
    return quant_int8_rows(x_bits, eps)


def per_token_cast_back(q: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """De-quantisation convention of the reference tests (float32-scale branch of
    tests/python/deepep/utils.py:182-188): bf16(float(q) * scale).  Returns bf16 bits.  uint8 payloads are E4M3 bit patterns."""
    qf = e4m3fn_bits_to_f32(q) if q.dtype == np.uint8 else q.astype(np.float32)
    return f32_to_bf16_bits_rne(qf * scale.astype(np.float32)[:, None])


# --------------------------------------------------------------------------------------
# A3  normal dispatch
# --------------------------------------------------------------------------------------
@dataclass
class DispatchResult:
    recv_x: np.ndarray                 # [max(R,1), H] int8 or bf16 bits
    recv_x_scales: Optional[np.ndarray]  # [max(R,1)] f32 (int8 mode)
    recv_src_idx: np.ndarray           # [3*max(R,1)] i32 triples (src_rank, token, k)
    num_recv_tokens_per_expert_list: List[int]
    send_head: np.ndarray              # [E] i32 == recv_count (round = 1)
    total_recv: int
    layout: dict = field(default_factory=dict)
    notify: dict = field(default_factory=dict)


def normal_dispatch(xs_bits: Sequence[np.ndarray], topk_idxs: Sequence[np.ndarray], num_experts: int,
                    quant: bool, expert_token_nums_type: int = 1) -> List[DispatchResult]:
    """All ranks' `Buffer.dispatch` results (default strategy, round = 1).

    Reference: host csrc/deepep/deep_ep.cpp:197-416; kernel cam_moe_dispatch_normal.h
      sender   :440-473  slot = send_data_offset[e] + send_token_idx_small[t,k] in own window,
                         row payload + scale + triple (src_rank, t, k) (:366-375)
      receiver :717-760  for idx i: count = recv_count[i] - recv_count[i-1]; rows
                         recv_offset[i] .. +count of rank src's window go to output rows
                         expert_global_offset[le] + srcrank_in_expert_offset[i] + r_in_srcrank_offset[i] + j
    => received rows are ordered (local expert, src rank, source row-major (t,k) order).
    Output sizes use max(R, 1) rows (deep_ep.cpp:327-328).  recv_topk_idx / recv_topk_weights are
    allocated but never written by the reference (deep_ep.cpp:371-374) and are not modelled.
    num_recv_tokens_per_expert_list: counts, or inclusive cumsum when MOE_EXPERT_TOKEN_NUMS_TYPE=0
    (deep_ep.cpp:384-401).
    """
    W = len(xs_bits)
    E = int(num_experts)
    L = E // W
    layouts = [dispatch_layout(topk_idxs[r], E, W) for r in range(W)]
    cnt = np.stack([l["num_tokens_per_expert"] for l in layouts]).astype(np.int64)
    Ts = [int(x.shape[0]) for x in xs_bits]
    H = int(xs_bits[0].shape[1])
    # ---- sender side: stage rows sorted by expert in the own window
    windows = []
    for r in range(W):
        x = np.ascontiguousarray(xs_bits[r]).view(np.uint16)
        ti = np.ascontiguousarray(topk_idxs[r], dtype=np.int64)
        T, K = ti.shape
        so = send_data_offset(cnt[r]).astype(np.int64)
        flat = ti.reshape(-1)
        vidx = np.nonzero((flat >= 0) & (flat < E))[0]
        slot = so[flat[vidx]] + layouts[r]["send_token_idx_small"].reshape(-1)[vidx]
        n = int(cnt[r].sum())
        tok = (vidx // K).astype(np.int32)
        kk = (vidx % K).astype(np.int32)
        if quant:
            q_all, s_all = _quant_rows(x, quant, 1e-12)
            payload = np.zeros((n, H), np.int8)
            payload[slot] = q_all[tok]
            scales = np.zeros(n, np.float32)
            scales[slot] = s_all[tok]
        else:
            payload = np.zeros((n, H), np.uint16)
            payload[slot] = x[tok]
            scales = None
        triple = np.zeros((n, 3), np.int32)
        triple[slot, 0] = r
        triple[slot, 1] = tok
        triple[slot, 2] = kk
        windows.append((payload, scales, triple))
    # ---- receiver side
    out = []
    for me in range(W):
        nt = notify_dispatch(cnt, Ts, me)
        R = int(nt["total_recv_token"])
        rows = max(R, 1)
        recv_x = np.zeros((rows, H), (np.uint8 if quant == "fp8" else np.int8) if quant else np.uint16)
        recv_s = np.zeros(rows, np.float32) if quant else None
        recv_t = np.zeros((rows, 3), np.int32)
        prev = 0
        for i in range(L * W):
            le, src = divmod(i, W)
            c = int(nt["recv_count"][i]) - prev
            prev = int(nt["recv_count"][i])
            if c == 0:
                continue
            so = int(nt["recv_offset"][i])
            dst = int(nt["expert_global_offset"][le]) + int(nt["srcrank_in_expert_offset"][i]) \
                + int(nt["r_in_srcrank_offset"][i])
            p, s, t3 = windows[src]
            recv_x[dst:dst + c] = p[so:so + c]
            if quant:
                recv_s[dst:dst + c] = s[so:so + c]
            recv_t[dst:dst + c] = t3[so:so + c]
        per_e = nt["recv_tokens_per_expert"].astype(np.int64)
        lst = (np.cumsum(per_e) if expert_token_nums_type == 0 else per_e).tolist()
        out.append(DispatchResult(recv_x, recv_s, recv_t.reshape(-1), [int(v) for v in lst],
                                  nt["recv_count"].copy(), R, layouts[me], nt))
    return out


# --------------------------------------------------------------------------------------
# A4 / A6  combine (normal and low-latency share the arithmetic)
# --------------------------------------------------------------------------------------
def weighted_reduce(rows_f32_by_k: np.ndarray, valid: np.ndarray, w: np.ndarray) -> np.ndarray:
    """acc = 0; for k ascending, valid only: acc = acc + (float(row_k) * w_k) with a separate fp32
    multiply and fp32 add (no FMA), then bf16 round-to-nearest-even.
    Reference: cam_moe_combine_normal.h:372-396 (Muls, Add, Cast CAST_RINT);
    moe_distribute_combine_v2.h:1102-1130,1192-1218,1252.
    rows_f32_by_k [T,K,H] f32, valid [T,K] bool, w [T,K] f32 -> bf16 bits [T,H]."""
    T, K, H = rows_f32_by_k.shape
    acc = np.zeros((T, H), np.float32)
    for k in range(K):
        prod = (rows_f32_by_k[:, k, :] * w[:, k].astype(np.float32)[:, None]).astype(np.float32)
        acc = np.where(valid[:, k][:, None], (acc + prod).astype(np.float32), acc)
    return f32_to_bf16_bits_rne(acc)


def combine(xs_bits: Sequence[np.ndarray], src_idx: Sequence[np.ndarray], total_rows: Sequence[int],
            topk_idxs: Sequence[np.ndarray], topk_weights: Sequence[Optional[np.ndarray]],
            num_experts: int) -> List[np.ndarray]:
    """All ranks' combine.  xs_bits[r] = expert-side rows [R_r, H] bf16 bits in dispatch order,
    src_idx[r] = the (src_rank, token, k) triples dispatch produced, total_rows[r] = R_r
    (= send_head[E-1], cam_moe_combine_normal.h:225).  topk_idxs / topk_weights are the ORIGINAL
    tensors of each source rank (handle[6], handle[7]; `combine(topk_weights=...)` is ignored by the
    reference, normal_strategy.py:407-420; None -> ones, deep_ep.cpp:568-572).

    Expert side pushes row r to rank `src`, slot t*K+k (cam_moe_combine_normal.h:291-321); the
    owner reduces the K slots of a token (weighted_reduce)."""
    W = len(xs_bits)
    H = int(xs_bits[0].shape[1])
    E = int(num_experts)
    slots = []
    for r in range(W):
        T, K = topk_idxs[r].shape
        slots.append(np.zeros((T * K, H), np.uint16))
    for r in range(W):
        n = int(total_rows[r])
        if n == 0:
            continue
        tri = np.asarray(src_idx[r], np.int32).reshape(-1, 3)[:n]
        x = np.ascontiguousarray(xs_bits[r]).view(np.uint16)[:n]
        for src in range(W):
            m = tri[:, 0] == src
            if not m.any():
                continue
            K = topk_idxs[src].shape[1]
            slots[src][tri[m, 1].astype(np.int64) * K + tri[m, 2]] = x[m]
    out = []
    for r in range(W):
        ti = np.asarray(topk_idxs[r], np.int64)
        T, K = ti.shape
        w = np.ones((T, K), np.float32) if topk_weights[r] is None else np.asarray(topk_weights[r], np.float32)
        valid = (ti >= 0) & (ti < E)
        rows = bf16_bits_to_f32(slots[r]).reshape(T, K, H)
        out.append(weighted_reduce(rows, valid, w))
    return out


# --------------------------------------------------------------------------------------
# A5  low-latency dispatch
# --------------------------------------------------------------------------------------
@dataclass
class LLDispatchResult:
    packed_recv_x: np.ndarray            # [M, H] int8 / bf16 bits; rows >= total are unspecified (0 here)
    packed_recv_x_scales: Optional[np.ndarray]  # [M] f32
    packed_recv_count: np.ndarray        # [L] int64
    src_info: np.ndarray                 # [3*total] meaningful prefix of the triples
    layout_range: np.ndarray             # [L*W] i32 inclusive cumsum over idx i
    total: int


def low_latency_dispatch(xs_bits: Sequence[np.ndarray], topk_idxs: Sequence[np.ndarray],
                         num_max_dispatch_tokens_per_rank: int, num_experts: int, quant: bool,
                         expert_token_nums_type: int = 1) -> List[LLDispatchResult]:
    """Reference: host csrc/deepep/deep_ep.cpp:850-1012; kernel moe_distribute_dispatch_v2.h
      sender :607-696   position among earlier (row-major) pairs of the same expert; row goes to the
                        destination rank's window region (src_rank, le) at that position
      counts :918-960   per (le, src) count + flag
      receiver :1253-1309  cumsum over idx i = le*W+src, rows packed back-to-back in that order;
                        layout_range (epRecvCounts) = inclusive cumsum; triples -> src_info
      counts out :1415-1455  packed_recv_count[le] = count (type 1) or cumulative (type 0), int64
    Output capacity M = W * max_tokens * min(K, L) (deep_ep.cpp:867-873).  INT8 quantisation as
    quant_int8_rows(eps=None)."""
    W = len(xs_bits)
    E = int(num_experts)
    L = E // W
    H = int(xs_bits[0].shape[1])
    K = int(topk_idxs[0].shape[1])
    M = W * int(num_max_dispatch_tokens_per_rank) * min(K, L)
    layouts = [dispatch_layout(np.asarray(topk_idxs[r], np.int64), E, W) for r in range(W)]
    cnt = np.stack([l["num_tokens_per_expert"] for l in layouts]).astype(np.int64)
    pre = []
    for r in range(W):
        x = np.ascontiguousarray(xs_bits[r]).view(np.uint16)
        pre.append(_quant_rows(x, quant, None) if quant else (x, None))
    out = []
    for me in range(W):
        rx = np.zeros((M, H), (np.uint8 if quant == "fp8" else np.int8) if quant else np.uint16)
        rs = np.zeros(M, np.float32) if quant else None
        tri = []
        rng = np.zeros(L * W, np.int32)
        pos = 0
        for le in range(L):
            e = me * L + le
            for src in range(W):
                ti = np.asarray(topk_idxs[src], np.int64)
                tt, kk = np.nonzero(ti == e)     # row-major order == sender position order
                c = tt.size
                assert c == int(cnt[src, e])
                if c:
                    rx[pos:pos + c] = pre[src][0][tt]
                    if quant:
                        rs[pos:pos + c] = pre[src][1][tt]
                    tri.append(np.stack([np.full(c, src, np.int32), tt.astype(np.int32), kk.astype(np.int32)], 1))
                pos += c
                rng[le * W + src] = pos
        per_e = cnt[:, me * L:(me + 1) * L].sum(axis=0)
        prc = (np.cumsum(per_e) if expert_token_nums_type == 0 else per_e).astype(np.int64)
        src_info = np.concatenate(tri).reshape(-1) if tri else np.zeros(0, np.int32)
        out.append(LLDispatchResult(rx, rs, prc, src_info, rng, pos))
    return out


# --------------------------------------------------------------------------------------
# A5 / A6 with shared-expert ranks (MOE_SHARED_EXPERT_RANK_NUM = S > 0).  PARITY UNPINNED: the reference's tests run this mode
# (tests/python/deepep/test_low_latency.py:385-389) but assert nothing about the shared ranks' rows; restated from the kernels.
# --------------------------------------------------------------------------------------
def low_latency_dispatch_shared(xs_bits: Sequence[np.ndarray], topk_idxs: Sequence[np.ndarray],
                                num_max_dispatch_tokens_per_rank: int, num_experts: int, quant, shared_expert_rank_num: int,
                                expert_token_nums_type: int = 1) -> List[LLDispatchResult]:
    """Reference: host csrc/deepep/deep_ep.cpp:866-874 (num_local_experts = 1 and M = global_bs / S on a shared rank, E / (W - S)
    experts and M = global_bs * min(K, L) on the others); kernel moe_distribute_dispatch_v2.h
      :555-604  SendToSharedExpert: every ACTIVE token (at least one selected expert, :748-779) of rank r goes to shared rank r mod S
                (one shared expert: rankNumPerSharedExpert = S), window region of source r, position = index among the active tokens,
                triple (token, k = K)
      :650-696  routed expert e lives on rank S + e / L (toRankId = dstExpertId / moeExpertNumPerRank + sharedExpertRankNum)
      :918-960  counts: a shared rank receives activeMaskBsCnt from every source with the same residue, nothing from the others."""
    W, S = len(xs_bits), int(shared_expert_rank_num)
    assert 0 < S < W and W % S == 0
    E = int(num_experts)
    L = E // (W - S)
    H = int(xs_bits[0].shape[1])
    K = int(topk_idxs[0].shape[1])
    MT = int(num_max_dispatch_tokens_per_rank)
    pre = []
    for r in range(W):
        x = np.ascontiguousarray(xs_bits[r]).view(np.uint16)
        pre.append(_quant_rows(x, quant, None) if quant else (x, None))
    out = []
    for me in range(W):
        shared = me < S
        nl = 1 if shared else L
        M = MT * W // S if shared else W * MT * min(K, L)
        rx = np.zeros((M, H), (np.uint8 if quant == "fp8" else np.int8) if quant else np.uint16)
        rs = np.zeros(M, np.float32) if quant else None
        tri, rng, per_e, pos = [], np.zeros(nl * W, np.int32), np.zeros(nl, np.int64), 0
        for le in range(nl):
            for src in range(W):
                ti = np.asarray(topk_idxs[src], np.int64)
                if shared:
                    active = ((ti >= 0) & (ti < E)).any(axis=1)
                    tt = np.nonzero(active)[0] if src % S == me else np.zeros(0, np.int64)
                    kk = np.full(tt.size, K, np.int64)
                else:
                    tt, kk = np.nonzero(ti == (me - S) * L + le)
                c = tt.size
                if c:
                    rx[pos:pos + c] = pre[src][0][tt]
                    if quant:
                        rs[pos:pos + c] = pre[src][1][tt]
                    tri.append(np.stack([np.full(c, src, np.int32), tt.astype(np.int32), kk.astype(np.int32)], 1))
                pos += c
                per_e[le] += c
                rng[le * W + src] = pos
        prc = (np.cumsum(per_e) if expert_token_nums_type == 0 else per_e).astype(np.int64)
        src_info = np.concatenate(tri).reshape(-1) if tri else np.zeros(0, np.int32)
        out.append(LLDispatchResult(rx, rs, prc, src_info, rng, pos))
    return out


def low_latency_combine_shared(xs_bits: Sequence[np.ndarray], src_idx: Sequence[np.ndarray], total_rows: Sequence[int],
                               topk_idxs: Sequence[np.ndarray], topk_weights: Sequence[np.ndarray], num_experts: int) -> List[np.ndarray]:
    """moe_distribute_combine_v2.h: every expert-side row goes to slot t * (K + 1) + k of its source rank (:885); the owner adds the K
    routed rows, each times its weight, k ascending (:1192-1218, as weighted_reduce), THEN the shared expert's row (slot K) unweighted
    (:1219-1235), and rounds to bf16 (:1252)."""
    W = len(xs_bits)
    H = int(xs_bits[0].shape[1])
    E = int(num_experts)
    slots = []
    for r in range(W):
        T, K = topk_idxs[r].shape
        slots.append(np.zeros((T * (K + 1), H), np.uint16))
    for r in range(W):
        n = int(total_rows[r])
        if n == 0:
            continue
        tri = np.asarray(src_idx[r], np.int32).reshape(-1, 3)[:n]
        x = np.ascontiguousarray(xs_bits[r]).view(np.uint16)[:n]
        for src in range(W):
            m = tri[:, 0] == src
            if m.any():
                K = topk_idxs[src].shape[1]
                slots[src][tri[m, 1].astype(np.int64) * (K + 1) + tri[m, 2]] = x[m]
    out = []
    for r in range(W):
        ti = np.asarray(topk_idxs[r], np.int64)
        T, K = ti.shape
        w = np.asarray(topk_weights[r], np.float32)
        valid = (ti >= 0) & (ti < E)
        rows = bf16_bits_to_f32(slots[r]).reshape(T, K + 1, H)
        acc = np.zeros((T, H), np.float32)
        for k in range(K):
            prod = (rows[:, k, :] * w[:, k][:, None]).astype(np.float32)
            acc = np.where(valid[:, k][:, None], (acc + prod).astype(np.float32), acc)
        acc = np.where(valid.any(axis=1)[:, None], (acc + rows[:, K, :]).astype(np.float32), acc)
        out.append(f32_to_bf16_bits_rne(acc))
    return out


# --------------------------------------------------------------------------------------
# closed-form goldens asserted by the reference tests (used to pin this oracle)
# --------------------------------------------------------------------------------------
def golden_combined(x_bits: np.ndarray, topk_idx: np.ndarray, topk_weights: np.ndarray) -> np.ndarray:
    """tests/python/deepep/test_intranode.py:431-436: x * sum_k w[k] * [topk_idx[k] != -1] (float)."""
    w = np.where(np.asarray(topk_idx) == -1, 0.0, np.asarray(topk_weights, np.float32)).sum(axis=1)
    return bf16_bits_to_f32(x_bits) * w.astype(np.float32)[:, None]


def calc_diff(x: np.ndarray, y: np.ndarray) -> float:
    """tests/python/deepep/utils.py:191-195."""
    x = x.astype(np.float64) + 1
    y = y.astype(np.float64) + 1
    return float(1 - 2 * (x * y).sum() / (x * x + y * y).sum())


# --------------------------------------------------------------------------------------
# A8  fused_deep_moe (dispatch -> INT8 grouped GEMM1 + SwiGLU -> requant -> GEMM2 -> combine)
# --------------------------------------------------------------------------------------
def permute_fusion_cols(n: int, tile: int = 128) -> np.ndarray:
    """Column permutation of reshape_fusion_gmm_weight / permute_weight (tests/python/deepep/test_fused_deep_moe.py:63-86):
    permuted column j*tile + h*(tile/2) + i  <-  original column h*(n/2) + j*(tile/2) + i  (h = 0 gate half, 1 up half)."""
    half = tile // 2
    j, h, i = np.meshgrid(np.arange(n // tile), np.arange(2), np.arange(half), indexing="ij")
    return (h * (n // 2) + j * half + i).reshape(-1)


def _int_matmul_exact(a_i8: np.ndarray, w_i8: np.ndarray) -> np.ndarray:
    """a [M, K] int8 x w [N, K] int8 -> int32 [M, N], exact.  Done in float64 through BLAS (|sum| <= K * 128 * 128 < 2^53 for any
    K this path allows, so every partial sum is an exactly representable integer); NumPy's integer matmul is a scalar loop."""
    assert a_i8.shape[1] * 128 * 128 < 2 ** 53
    return np.rint(a_i8.astype(np.float64) @ w_i8.astype(np.float64).T).astype(np.int32)


def moe_gemm1_swiglu(a_i8: np.ndarray, a_scale: np.ndarray, w_i8: np.ndarray, w_scale: np.ndarray) -> np.ndarray:
    """One expert's GEMM1 + per-token dequant + SwiGLU.  a [rows, H] int8 with per-token scales, w [2I, H] int8 / w_scale [2I]
    in ORIGINAL column order (first I gate, last I up).  d = (float(c) * w_scale[col]) * tok_scale[row]
    (block_epilogue_per_token_dequant_swiglu.h:250-269); v = up * gate / (1 + exp(-gate)).  -> f32 [rows, I]."""
    c = _int_matmul_exact(a_i8, w_i8)                                                     # [rows, 2I] exact
    dd = (c.astype(np.float32) * np.asarray(w_scale, np.float32)[None, :]) * np.asarray(a_scale, np.float32)[:, None]
    I = dd.shape[1] // 2
    gate, up = dd[:, :I], dd[:, I:]
    with np.errstate(over="ignore"):
        return (up * (gate / (np.float32(1.0) + np.exp(-gate).astype(np.float32)))).astype(np.float32)


def moe_rowquant(v: np.ndarray):
    """Per-row symmetric requantisation of the SwiGLU output: q = rint((v * 127) * (1 / rowmax)), scale = rowmax / 127
    (...grouped_matmul_slice_m_per_token_dequant_swiglu_quant_multistage_workspace.h:199-265).  -> (int8 [rows, I], f32 [rows])."""
    v = np.asarray(v, np.float32)
    rowmax = np.abs(v).max(axis=1).astype(np.float32)
    inv = np.where(rowmax > 0, np.float32(1.0) / np.where(rowmax > 0, rowmax, 1), 0).astype(np.float32)
    q = np.rint((v * np.float32(127.0)) * inv[:, None]).astype(np.int32)
    return q.astype(np.int8), (rowmax / np.float32(127.0)).astype(np.float32)


def moe_gemm2(q_i8: np.ndarray, q_scale: np.ndarray, w_i8: np.ndarray, w_scale: np.ndarray) -> np.ndarray:
    """One expert's GEMM2 + per-token x per-channel dequant: bf16((float(c2) * w_scale[col]) * scale[row]).  w [H, I] int8.
    -> bf16 bits [rows, H]."""
    c2 = _int_matmul_exact(q_i8, w_i8)
    out = (c2.astype(np.float32) * np.asarray(w_scale, np.float32)[None, :]) * np.asarray(q_scale, np.float32)[:, None]
    return f32_to_bf16_bits_rne(out)


def fused_deep_moe(xs_bits, topk_idxs, topk_weights, w13, w13_scale, w2, w2_scale, num_max_dispatch_tokens_per_rank,
                   num_experts, shared_expert_rank_num: int = 0) -> List[np.ndarray]:
    """All ranks' fused_deep_moe.  w13[r] int8 [L, 2I, H] / w13_scale[r] f32 [L, 2I] in ORIGINAL column order (first half gate,
    second half up; the permutation only changes where a column sits, not the math), w2[r] int8 [L, H, I], w2_scale[r] f32 [L, H].
    Arithmetic: INT8 per-token dispatch without epsilon; d = (float(c) * w_scale[col]) * tok_scale[row]
    (block_epilogue_per_token_dequant_swiglu.h:250-269); v = up * gate / (1 + exp(-gate)); q = rint((v*127) * (1/rowmax)),
    scale = rowmax/127 (...swiglu_quant_multistage_workspace.h:199-265); y = bf16((float(c2) * w2_scale[col]) * scale[row]);
    combine as A6.  Returns bf16 bits per rank.  shared_expert_rank_num = S > 0 (deep_ep.cpp:1219-1220): ranks < S hold one expert
    (w13[r] has L = 1), the others num_experts / (W - S); dispatch / combine as low_latency_dispatch_shared / low_latency_combine_shared."""
    W = len(xs_bits)
    E = int(num_experts)
    S = int(shared_expert_rank_num)
    if S:
        disp = low_latency_dispatch_shared(xs_bits, topk_idxs, num_max_dispatch_tokens_per_rank, E, True, S)
    else:
        disp = low_latency_dispatch(xs_bits, topk_idxs, num_max_dispatch_tokens_per_rank, E, quant=True)
    ys = []
    for r in range(W):
        d = disp[r]
        L = int(w13[r].shape[0])
        H = d.packed_recv_x.shape[1]
        y = np.zeros((d.packed_recv_x.shape[0], H), np.uint16)
        start = 0
        for le in range(L):
            end = int(d.layout_range[(le + 1) * W - 1])
            if end > start:
                v = moe_gemm1_swiglu(d.packed_recv_x[start:end], d.packed_recv_x_scales[start:end], w13[r][le], w13_scale[r][le])
                q, sc = moe_rowquant(v)
                y[start:end] = moe_gemm2(q, sc, w2[r][le], w2_scale[r][le])
            start = end
        ys.append(y)
    if S:
        return low_latency_combine_shared(ys, [d.src_info for d in disp], [d.total for d in disp], topk_idxs, topk_weights, E)
    return combine(ys, [d.src_info for d in disp], [d.total for d in disp], topk_idxs, topk_weights, E)
