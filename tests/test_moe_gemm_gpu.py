"""GPU parity tests of the grouped INT8 expert GEMMs of fused_deep_moe, driven through the C-ABI (include/mi_ep.h:
mi_ep_moe_gemm1_swiglu / mi_ep_moe_rowquant / mi_ep_moe_gemm2 / mi_ep_moe_gemm2_push) with ctypes.

Both workgroup-tile variants are exercised: rows_per_expert_hint <= 96 selects the 64-row tile (decode-size groups), anything
else the 256 x 256 x 64 tile that produces the BASELINE C5 number.  Checks:
  * small / medium shapes against the CPU oracle (oracle/ep.py: moe_gemm1_swiglu, moe_rowquant, moe_gemm2 -- the arithmetic of
    the reference's epilogues, block_epilogue_per_token_dequant_swiglu.h:250-269 and
    ...swiglu_quant_multistage_workspace.h:199-265);
  * DeepSeek-V3 sized shapes (H = 7168, 2I = 4096, 32 local experts, ragged counts) against the same arithmetic restated with
    torch on the GPU: the int32 accumulators are exact in fp64 (|c| <= 7168 * 127 * 128 < 2^53), so GEMM2 and the requant are
    compared BIT-EXACT and GEMM1 within the accuracy of the fast exponential (the reference test's own bar is avg_diff < 4e-4,
    tests/python/deepep/test_fused_deep_moe.py:470)."""
import ctypes
from ctypes import c_int, c_void_p

import numpy as np
import pytest
import torch

from capi import load, ptr, ptr_array, stream_ptr
from oracle import ep as O
from oracle.bf16 import bits_to_torch, torch_to_bits

pytestmark = pytest.mark.gpu

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        L = load("libmi_ep.so")
        V, I = c_void_p, c_int
        L.mi_ep_moe_gemm1_swiglu.argtypes = [V, V, V, V, V, I, I, I, I, I, V, I, V]
        L.mi_ep_moe_rowquant.argtypes = [V, V, I, I, V, V, V]
        L.mi_ep_moe_gemm2.argtypes = [V, V, V, V, V, I, I, I, I, I, V, I, V]
        L.mi_ep_moe_gemm2_push.argtypes = [V, V, V, V, V, I, I, I, I, I, V, I, V, I, ctypes.c_size_t, V, ctypes.c_size_t, I, V]
        L.mi_ep_combine_row_bytes.restype = ctypes.c_size_t
        L.mi_ep_combine_row_bytes.argtypes = [I]
        L.mi_ep_moe_gemm1_swiglu_quant.argtypes = [V, V, V, V, V, V, I, I, I, I, I, V, V, V, I, V, I, I, V]
        L.mi_ep_moe_requant_words.restype = ctypes.c_size_t
        L.mi_ep_moe_requant_words.argtypes = [I, I]
        L.mi_ep_moe_probe_xcds.argtypes = [V]
        for n in ("mi_ep_moe_gemm1_swiglu", "mi_ep_moe_rowquant", "mi_ep_moe_gemm2", "mi_ep_moe_gemm2_push", "mi_ep_moe_gemm1_swiglu_quant",
                  "mi_ep_moe_probe_xcds"):
            getattr(L, n).restype = c_int
        _LIB = L
    return _LIB


def ck(rc):
    assert rc == 0, f"mi_ep call failed rc={rc}"


def make_cum(counts, stride, dev):
    """Inclusive cumulative table with `stride` entries per expert (layout_range has W per expert: only the last one of an
    expert is its end; the others are intermediate prefixes like the per-source entries of a real layout_range)."""
    ends = np.cumsum(counts)
    cum = np.zeros(len(counts) * stride, np.int32)
    start = 0
    for e, end in enumerate(ends):
        # intermediate entries: any non-decreasing split of the expert's rows
        parts = np.linspace(start, end, stride + 1)[1:].astype(np.int32)
        parts[-1] = end
        cum[e * stride:(e + 1) * stride] = parts
        start = end
    return torch.from_numpy(cum).to(dev)


def fusion_perm(two_i):
    return O.permute_fusion_cols(two_i)


def run_gemm1(a, a_scale, w_perm, ws_perm, cum, stride, L, rows_cap, H, two_i, hint):
    out = torch.full((rows_cap, two_i // 2), float("nan"), dtype=torch.float32, device=a.device)
    ck(lib().mi_ep_moe_gemm1_swiglu(ptr(a), ptr(a_scale), ptr(w_perm), ptr(ws_perm), ptr(cum), stride, L, rows_cap, H, two_i,
                                    ptr(out), hint, stream_ptr()))
    return out


def run_rowquant(v, total_dev, rows_cap, I):
    q = torch.zeros((rows_cap, I), dtype=torch.int8, device=v.device)
    sc = torch.zeros(rows_cap, dtype=torch.float32, device=v.device)
    ck(lib().mi_ep_moe_rowquant(ptr(v), ptr(total_dev), rows_cap, I, ptr(q), ptr(sc), stream_ptr()))
    return q, sc


def run_gemm1_quant(a, a_scale, w_perm, ws_perm, cum, stride, L, rows_cap, H, two_i, hint, xcds=None, row_offsets=None):
    """GEMM1 with the requantisation in its epilogue (mi_ep_moe_gemm1_swiglu_quant): q, scale and the status word."""
    Lb = lib()
    q = torch.zeros((rows_cap, two_i // 2), dtype=torch.int8, device=a.device)
    sc = torch.zeros(rows_cap, dtype=torch.float32, device=a.device)
    words = torch.zeros(Lb.mi_ep_moe_requant_words(rows_cap, L), dtype=torch.int32, device=a.device)      # (zero at launch: the contract)
    status = torch.zeros(4, dtype=torch.int32, device=a.device)
    if xcds is None:
        xcds = Lb.mi_ep_moe_probe_xcds(stream_ptr())
    ck(Lb.mi_ep_moe_gemm1_swiglu_quant(ptr(a), ptr(row_offsets) if row_offsets is not None else None, ptr(a_scale), ptr(w_perm), ptr(ws_perm),
                                       ptr(cum), stride, L, rows_cap, H, two_i, ptr(q), ptr(sc), ptr(words), xcds, ptr(status), 5000, hint,
                                       stream_ptr()))
    return q, sc, status


def run_gemm2(q, sc, w2, s2, cum, stride, L, rows_cap, I, H, hint):
    out = torch.zeros((rows_cap, H), dtype=torch.bfloat16, device=q.device)
    ck(lib().mi_ep_moe_gemm2(ptr(q), ptr(sc), ptr(w2), ptr(s2), ptr(cum), stride, L, rows_cap, I, H, ptr(out), hint, stream_ptr()))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# vs the CPU oracle
# ---------------------------------------------------------------------------------------------------------------------
ORACLE_CASES = [
    # counts per expert, H, I, cum stride, hint
    ([5, 0, 1, 70], 256, 128, 1, 16),             # 64-row tile, 0-row and 1-row experts
    ([5, 0, 1, 70], 256, 128, 2, 0),              # same groups on the 256-row tile (hint 0 = unknown)
    ([300, 1, 0, 257, 256, 511], 512, 256, 1, 512),   # 256-row tile: counts around the tile size, not multiples of 256
    ([300, 1, 0, 257, 256, 511], 512, 256, 4, 64),    # same on the 64-row tile
    ([130] * 3 + [0] * 70 + [3, 260], 128, 128, 1, 200),   # > 64 experts: the tile lookup needs its second 64-expert step
    ([97, 33], 1024, 384, 1, 97),                 # 2I = 768 = 3 x 256-column tiles, I not a multiple of 256
    ([5, 0, 1, 70, 33], 256, 192, 1, 16),         # GEMM2's K = 192 is no multiple of 128: 64-row tile with 64-byte k-tiles
    ([300, 1, 257], 256, 192, 1, 512),            # the same K on the 256-row tile (GEMM1: 128-byte k-tiles, GEMM2: 64-byte k-tiles)
    # the 256-row kernel runs an expert's last row block of <= 64 / <= 128 rows as a 64- / 128-row tile: every class of remainder, at the edges
    ([356, 65, 128, 129, 64, 320, 0, 577], 512, 256, 1, 512),
    ([1040, 1000, 1088, 1089, 1024], 1024, 512, 2, 1024),
]


@pytest.mark.parametrize("counts,H,I,stride,hint", ORACLE_CASES)
def test_moe_gemm_chain_vs_oracle(counts, H, I, stride, hint):
    dev = torch.device("cuda")
    rng = np.random.default_rng(len(counts) * 1000 + H + hint)
    L, total = len(counts), int(sum(counts))
    rows_cap = total + 37                          # capacity beyond the valid rows (never touched)
    a = rng.integers(-127, 128, (rows_cap, H)).astype(np.int8)
    a_scale = (rng.random(rows_cap) * 0.02 + 0.005).astype(np.float32)
    w13 = rng.integers(-16, 16, (L, 2 * I, H)).astype(np.int8)            # ORIGINAL column order (gate | up)
    s13 = (rng.random((L, 2 * I)) * 4e-4 + 1.5e-3).astype(np.float32)
    w2 = rng.integers(-16, 16, (L, H, I)).astype(np.int8)
    s2 = (rng.random((L, H)) * 4e-4 + 1.5e-3).astype(np.float32)
    perm = fusion_perm(2 * I)
    cum = make_cum(counts, stride, dev)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    v = run_gemm1(t(a), t(a_scale), t(w13[:, perm, :]), t(s13[:, perm]), cum, stride, L, rows_cap, H, 2 * I, hint)
    total_dev = torch.tensor([total], dtype=torch.int32, device=dev)
    q, sc = run_rowquant(v, total_dev, rows_cap, I)
    y = run_gemm2(q, sc, t(w2), t(s2), cum, stride, L, rows_cap, I, H, hint)
    if (2 * I) % 256 == 0:
        # the requantisation in GEMM1's epilogue: the bits of the two launches, with the probed XCD count and with "no placement assumed",
        # from the dense rows and through a row-offset table (rows 128-byte aligned, in another order)
        for xcds in (None, 1, 2, 4):        # (2, 4: the ticket groups a partitioned device would form; the launcher falls back to 1 when gx % xcds != 0)
            q_f, sc_f, st_f = run_gemm1_quant(t(a), t(a_scale), t(w13[:, perm, :]), t(s13[:, perm]), cum, stride, L, rows_cap, H, 2 * I, hint, xcds)
            assert int(st_f[0]) == 0
            assert torch.equal(q_f[:total], q[:total]) and torch.equal(sc_f[:total].view(torch.int32), sc[:total].view(torch.int32)), xcds
            assert not bool(q_f[total:].any()) and not bool(sc_f[total:].any()), "rows past the last expert were written"
        stride_b = (H + 127) // 128 * 128 + 128
        order = rng.permutation(rows_cap)
        scattered = np.zeros((rows_cap, stride_b), np.int8)
        scattered[order, :H] = a
        offs = t((order.astype(np.int64) * stride_b).astype(np.uint32).view(np.int32))
        q_r, sc_r, st_r = run_gemm1_quant(t(scattered), t(a_scale), t(w13[:, perm, :]), t(s13[:, perm]), cum, stride, L, rows_cap, H, 2 * I, hint,
                                          None, offs)
        assert int(st_r[0]) == 0 and torch.equal(q_r[:total], q[:total]) and torch.equal(sc_r[:total].view(torch.int32), sc[:total].view(torch.int32))
    torch.cuda.synchronize()
    v_h, q_h, sc_h, y_h = v.cpu().numpy(), q.cpu().numpy(), sc.cpu().numpy(), torch_to_bits(y)
    assert np.isnan(v_h[total:]).all(), "rows past the last expert were written"
    start = 0
    for e, c in enumerate(counts):
        if c == 0:
            continue
        sl = slice(start, start + c)
        v_want = O.moe_gemm1_swiglu(a[sl], a_scale[sl], w13[e], s13[e])
        # fast exponential in the kernel vs libm in the oracle: a few ulp on the sigmoid
        np.testing.assert_allclose(v_h[sl], v_want, rtol=3e-5, atol=1e-6 * np.abs(v_want).max())
        # the stages after GEMM1 are exact functions of the kernel's own v
        q_want, sc_want = O.moe_rowquant(v_h[sl])
        assert np.array_equal(q_h[sl], q_want), e
        assert np.array_equal(sc_h[sl].view(np.uint32), sc_want.view(np.uint32)), e
        y_want = O.moe_gemm2(q_h[sl], sc_h[sl], w2[e], s2[e])
        assert np.array_equal(y_h[sl], y_want), e
        start += c
    assert not y_h[total:].any(), "GEMM2 wrote rows past the last expert"


# ---------------------------------------------------------------------------------------------------------------------
# DeepSeek-V3 sizes (the C5 shapes): torch-on-GPU restatement of the same arithmetic, exact accumulators
# ---------------------------------------------------------------------------------------------------------------------
def torch_gemm1(a, a_scale, w_perm, ws_perm):
    c = (a.double() @ w_perm.double().T)                               # exact: |c| < 2^53
    d = (c.float() * ws_perm[None, :]) * a_scale[:, None]
    d = d.view(d.shape[0], -1, 2, 64)                                  # fusion tiles: 64 gate | 64 up
    gate, up = d[:, :, 0, :], d[:, :, 1, :]
    return (up * (gate / (1.0 + torch.exp(-gate)))).reshape(d.shape[0], -1)


def torch_rowquant(v):
    rowmax = v.abs().amax(dim=1)
    inv = torch.where(rowmax > 0, 1.0 / rowmax, torch.zeros_like(rowmax))
    q = torch.round((v * 127.0) * inv[:, None]).to(torch.int8)
    # tensor / tensor: a scalar divisor would be turned into a multiplication by its reciprocal by the elementwise kernel
    return q, rowmax / torch.full_like(rowmax, 127.0)


def torch_gemm2(q, sc, w2, s2):
    c2 = (q.double() @ w2.double().T)
    return ((c2.float() * s2[None, :]) * sc[:, None]).to(torch.bfloat16)


def c5_counts(rng, L, avg):
    counts = rng.integers(avg // 2, avg * 3 // 2, L)
    counts[3], counts[7], counts[11] = 0, 1, 255                       # empty, single-row and just-under-a-tile experts
    counts[12], counts[13] = 256, 257
    return [int(c) for c in counts]


@pytest.mark.parametrize("hint", [0, 1024, 64], ids=["tile256_unknown", "tile256", "tile64"])
def test_moe_gemm_c5_shapes_exact(hint):
    """H = 7168, 2I = 4096, 32 local experts (BASELINE C5, EP = 8): ~700 rows per expert for the 256-row tile (> 64 tiles),
    ~40 for the 64-row tile."""
    dev = torch.device("cuda")
    H, I, L = 7168, 2048, 32
    rng = np.random.default_rng(11 + hint)
    counts = c5_counts(rng, L, 700 if hint != 64 else 40)
    total = sum(counts)
    rows_cap = total + 300
    g = torch.Generator(device="cuda").manual_seed(5 + hint)
    ri = lambda lo, hi, shape: torch.randint(lo, hi, shape, generator=g, device=dev, dtype=torch.int32).to(torch.int8)
    a = ri(-127, 128, (rows_cap, H))
    a_scale = torch.rand(rows_cap, generator=g, device=dev) * 0.02 + 0.005
    w13p = ri(-16, 16, (L, 2 * I, H))                                   # already in fusion-tile order
    s13p = torch.rand((L, 2 * I), generator=g, device=dev) * 4e-4 + 1.5e-3
    w2 = ri(-16, 16, (L, H, I))
    s2 = torch.rand((L, H), generator=g, device=dev) * 4e-4 + 1.5e-3
    stride = 8
    cum = make_cum(counts, stride, dev)
    v = run_gemm1(a, a_scale, w13p, s13p, cum, stride, L, rows_cap, H, 2 * I, hint)
    total_dev = torch.tensor([total], dtype=torch.int32, device=dev)
    q, sc = run_rowquant(v, total_dev, rows_cap, I)
    y = run_gemm2(q, sc, w2, s2, cum, stride, L, rows_cap, I, H, hint)
    for rep in range(3):        # GEMM1 with the requantisation in its epilogue: the same bits, every time (the workers form at run time)
        q_f, sc_f, st_f = run_gemm1_quant(a, a_scale, w13p, s13p, cum, stride, L, rows_cap, H, 2 * I, hint)
        assert int(st_f[0]) == 0
        assert torch.equal(q_f[:total], q[:total]) and torch.equal(sc_f[:total].view(torch.int32), sc[:total].view(torch.int32)), rep
        assert not bool(q_f[total:].any())
    torch.cuda.synchronize()
    assert bool(torch.isnan(v[total:]).all())
    assert not bool(y[total:].any())
    start = 0
    sum_abs, sum_ref = 0.0, 0.0
    for e, c in enumerate(counts):
        if c == 0:
            continue
        sl = slice(start, start + c)
        v_want = torch_gemm1(a[sl], a_scale[sl], w13p[e], s13p[e])
        err = (v[sl] - v_want).abs()
        assert bool((err <= 3e-5 * v_want.abs() + 1e-6 * v_want.abs().max()).all()), (e, err.max().item())
        sum_abs += err.sum().item()
        sum_ref += v_want.abs().sum().item()
        q_want, sc_want = torch_rowquant(v[sl])
        assert torch.equal(q[sl], q_want), e
        assert torch.equal(sc[sl].view(torch.int32), sc_want.view(torch.int32)), e
        y_want = torch_gemm2(q[sl], sc[sl], w2[e], s2[e])
        assert torch.equal(y[sl].view(torch.int16), y_want.view(torch.int16)), e     # bit-exact: int32 accumulators are exact
        start += c
    assert sum_abs / max(sum_ref, 1e-30) < 1e-5


@pytest.mark.parametrize("hint", [1024, 64], ids=["tile256", "tile64"])
def test_moe_gemm2_push_lands_in_combine_slots(hint):
    """mi_ep_moe_gemm2_push == mi_ep_moe_gemm2 followed by mi_ep_combine_push: row r goes to slot t*K+k of rank src."""
    dev = torch.device("cuda")
    H, I, L, W, K = 1024, 512, 6, 3, 4
    rng = np.random.default_rng(99 + hint)
    counts = [300, 0, 1, 257, 90, 513] if hint != 64 else [30, 0, 1, 65, 9, 64]
    total = sum(counts)
    rows_cap = total + 11
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    q = t(rng.integers(-127, 128, (rows_cap, I)).astype(np.int8))
    sc = t((rng.random(rows_cap) * 0.01 + 0.001).astype(np.float32))
    w2 = t(rng.integers(-16, 16, (L, H, I)).astype(np.int8))
    s2 = t((rng.random((L, H)) * 4e-4 + 1.5e-3).astype(np.float32))
    cum = make_cum(counts, 1, dev)
    # every valid row gets a distinct (src, t, k) slot
    T = (total + W * K - 1) // (W * K) + 2
    slots = rng.permutation(W * T * K)[:total]
    tri = np.zeros((rows_cap, 3), np.int32)
    tri[:total, 0], tri[:total, 1], tri[:total, 2] = slots // (T * K), (slots // K) % T, slots % K
    tri[total:] = -7                                                    # garbage past the valid rows must never be used
    src_idx = t(tri.reshape(-1))
    cb = lib().mi_ep_combine_row_bytes(H)
    wins = [torch.zeros(T * K * cb, dtype=torch.uint8, device=dev) for _ in range(W)]
    ck(lib().mi_ep_moe_gemm2_push(ptr(q), ptr(sc), ptr(w2), ptr(s2), ptr(cum), 1, L, rows_cap, I, H, ptr(src_idx), K,
                                  ptr_array([w.data_ptr() for w in wins]), W, wins[0].numel(), None, 0, hint, stream_ptr()))
    dense = run_gemm2(q, sc, w2, s2, cum, 1, L, rows_cap, I, H, hint)
    torch.cuda.synchronize()
    want = [torch.zeros((T * K, cb // 2), dtype=torch.int16, device=dev) for _ in range(W)]
    dv = dense.view(torch.int16)
    for r in range(total):
        want[tri[r, 0]][tri[r, 1] * K + tri[r, 2], :H] = dv[r]
    for s in range(W):
        assert torch.equal(wins[s].view(torch.int16).view(T * K, cb // 2), want[s]), s
