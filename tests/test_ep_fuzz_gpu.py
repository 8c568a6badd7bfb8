"""Seeded random sweep of the dispatch / combine kernels (through the C-ABI, W ranks in one process) against the CPU oracle,
bit-exact: odd hidden sizes (any multiple of 16 up to 8192), top-k up to 16, up to 2048 experts, ragged and empty ranks, heavy
drop rates, all three normal-dispatch forms, both quantisation modes, low-latency mode.  The fixed cases of test_ep_gpu.py pin the
BASELINE shapes; this one looks for corner cases between them."""
import numpy as np
import pytest
import torch

from oracle import ep as O
from oracle.bf16 import torch_to_bits
from test_ep_gpu import dev_bf16, make_topk, rand_bits

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    W = int(rng.choice([1, 2, 3, 4, 5, 8]))
    L = int(rng.choice([1, 2, 3, 8, 32, 64]))
    if rng.random() < 0.15:
        L = 2048 // W // (1 if W in (1, 2, 4, 8) else 2)          # up to the 2048-expert limit
    E = L * W
    K = int(rng.integers(1, min(16, E) + 1))
    H = int(rng.integers(1, 513)) * 16
    T = int(rng.integers(0, 200))
    drop = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
    return rng, W, E, K, H, T, drop


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("mode", ["replicated", "compact", "push"])
def test_random_normal_dispatch_combine(seed, mode):
    import ep_harness as Hh
    rng, W, E, K, H, T, drop = _case(1000 + seed)
    quant = bool(seed % 2)
    Ts = [max(0, T + int(rng.integers(-T // 2 - 1, 20))) for _ in range(W)]
    if W > 1 and seed % 3 == 0:
        Ts[int(rng.integers(0, W))] = 0                              # an empty rank
    xs = [rand_bits(rng, (t, H), float(rng.choice([0.01, 1.0, 100.0]))) for t in Ts]
    idxs = [make_topk(rng, t, K, E, drop) if t else np.zeros((0, K), np.int64) for t in Ts]
    ws = [rng.standard_normal((t, K)).astype(np.float32) for t in Ts]
    h = Hh.InProcEP(W, E, max(Ts) + 1, K, H, compact=mode == "compact", transport="push" if mode == "push" else "pull")
    qm = Hh.QUANT_INT8 if quant else Hh.QUANT_NONE
    to_idx = (lambda a: torch.from_numpy(a).int().cuda()) if seed % 4 == 1 else (lambda a: torch.from_numpy(a).cuda())
    got = h.dispatch([dev_bf16(x) for x in xs], [to_idx(i) for i in idxs], qm)
    want = O.normal_dispatch(xs, idxs, E, quant)
    for r in range(W):
        g, w = got[r], want[r]
        n = w.total_recv
        assert g["total"] == n, (r, g["total"], n)
        assert np.array_equal(g["tables"]["recv_count"].cpu().numpy().reshape(-1), np.asarray(w.notify["recv_count"]).reshape(-1))
        assert np.array_equal(g["recv_src_idx"].cpu().numpy()[:3 * n], w.recv_src_idx[:3 * n])
        if quant:
            assert np.array_equal(g["recv_x"].cpu().numpy()[:n], w.recv_x[:n])
            assert np.array_equal(g["recv_x_scales"].cpu().numpy()[:n].view(np.uint32), w.recv_x_scales[:n].view(np.uint32))
        else:
            assert np.array_equal(torch_to_bits(g["recv_x"])[:n], w.recv_x[:n])
    ys = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
    comb_want = O.combine(ys, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
    # push mode: the dispatch already knows the receive row of every own-rank selection (pull_local's local_row table): at W = 1 the combine
    # is the reduce alone, at W > 1 the table must equal the one the combine push builds
    comb_got = h.combine([dev_bf16(y) for y in ys], [g["recv_src_idx"] for g in got], [g["total"] for g in got],
                         [to_idx(i) for i in idxs], [torch.from_numpy(w_).cuda() for w_ in ws],
                         dispatch_local_rows=[g["local_row"] for g in got] if mode == "push" else None)
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), r


@pytest.mark.parametrize("seed", range(16))
def test_random_low_latency(seed):
    import ep_harness as Hh
    rng, W, E, K, H, T, drop = _case(5000 + seed)
    L = E // W
    if L * W > 2048 or L * W * max(T, 1) * (H * 2 + 16) > (1 << 30):      # keep the slab window of the harness small
        E = W * min(L, 8)
    T = max(1, min(T, 64))
    quant = bool(seed % 2)
    Ts = [max(1, T - int(rng.integers(0, 3))) for _ in range(W)]
    xs = [rand_bits(rng, (t, H), 2.0) for t in Ts]
    K = min(K, E)
    idxs = [make_topk(rng, t, K, E, drop) for t in Ts]
    ws = [np.abs(rng.standard_normal((t, K))).astype(np.float32) for t in Ts]
    h = Hh.InProcEP(W, E, T, K, H)
    qm = Hh.QUANT_INT8_NOEPS if quant else Hh.QUANT_NONE
    ct = int(seed % 3 != 0)
    got = h.ll_dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).int().cuda() for i in idxs], qm, ct)
    want = O.low_latency_dispatch(xs, idxs, T, E, quant, expert_token_nums_type=ct)
    for r in range(W):
        g, w = got[r], want[r]
        n = w.total
        assert np.array_equal(g["layout_range"].cpu().numpy(), w.layout_range)
        assert np.array_equal(g["packed_recv_count"].cpu().numpy(), w.packed_recv_count)
        assert np.array_equal(g["src_info"].cpu().numpy()[:3 * n], w.src_info)
        if quant:
            assert np.array_equal(g["packed_recv_x"].cpu().numpy()[:n], w.packed_recv_x[:n])
            assert np.array_equal(g["packed_recv_x_scales"].cpu().numpy()[:n].view(np.uint32), w.packed_recv_x_scales[:n].view(np.uint32))
        else:
            assert np.array_equal(torch_to_bits(g["packed_recv_x"])[:n], w.packed_recv_x[:n])
    ys = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) if quant else w.packed_recv_x for w in want]
    comb_want = O.combine(ys, [w.src_info for w in want], [w.total for w in want], idxs, ws, E)
    comb_got = h.combine([dev_bf16(y) for y in ys], [g["src_info"] for g in got], [w.total for w in want],
                         [torch.from_numpy(i).int().cuda() for i in idxs], [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), r


@pytest.mark.parametrize("seed", range(12))
def test_random_alltoall_transport_with_absent_selections(seed):
    """The all-to-all (RCCL fallback) kernels under heavy drop rates: the return buffer of the gather-mode reduce holds exactly the
    VALID pairs, so an absent selection (-1 or out of range) must neither be summed nor addressed past the end of that buffer
    (late tokens' slot t*K lies beyond it as soon as anything was dropped)."""
    import ep_harness as Hh
    rng, W, E, K, H, T, _ = _case(9000 + seed)
    T = max(T, 40)
    drop = float(rng.choice([0.1, 0.5, 0.9]))
    quant = bool(seed % 2)
    Ts = [T + r for r in range(W)]
    xs = [rand_bits(rng, (t, H), 1.0) for t in Ts]
    idxs = [make_topk(rng, t, K, E, drop) for t in Ts]
    for i in idxs:                                                       # out-of-range ids are absent selections as well
        i[rng.random(i.shape) < 0.05] = E + int(rng.integers(0, 5))
    if seed % 4 == 0:
        idxs[0][:] = -1                                                  # a rank that sends nothing: its return buffer is the 1-row dummy
    ws = [rng.standard_normal((t, K)).astype(np.float32) for t in Ts]
    a2a = Hh.InProcA2A(W, E, K, H)
    qm = Hh.QUANT_INT8 if quant else Hh.QUANT_NONE
    got = a2a.dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).cuda() for i in idxs], qm)
    want = O.normal_dispatch(xs, idxs, E, quant)
    for r in range(W):
        n = want[r].total_recv
        assert got[r]["total"] == n
        assert np.array_equal(got[r]["recv_src_idx"].cpu().numpy()[:3 * n], want[r].recv_src_idx[:3 * n])
    ys = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
    comb_want = O.combine(ys, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
    comb_got = a2a.combine([dev_bf16(y) for y in ys], got, [torch.from_numpy(i).cuda() for i in idxs],
                           [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), r


@pytest.mark.parametrize("mode", ["replicated", "compact", "push"])
@pytest.mark.parametrize("eps", [True, False])
def test_non_finite_rows_quantise_as_the_oracle_says(mode, eps):
    """A row holding a NaN or an infinity: max |x| = +inf (the NaN pattern is clamped to infinity in the packed integer maximum) ->
    scale_out = +inf and q = 0 at every finite element, in both quantisation modes; the other rows are untouched by it.  The bytes
    AT the non-finite positions are unspecified (oracle/ep.py quant_int8_rows) and not compared."""
    import ep_harness as Hh
    rng = np.random.default_rng(77)
    W, E, K, H, T = 2, 8, 2, 1024, 24
    xs = [rand_bits(rng, (T, H), 1.0) for _ in range(W)]
    bad = {}
    for r in range(W):
        xs[r][3, 17] = 0x7FC0          # NaN
        xs[r][5, 900] = 0xFFFF         # NaN, sign set, all mantissa bits
        xs[r][9, 0] = 0x7F80           # +inf
        xs[r][11, 513] = 0xFF80        # -inf
        bad[r] = {3: 17, 5: 900, 9: 0, 11: 513}
    idxs = [make_topk(rng, T, K, E, 0.0) for _ in range(W)]
    if eps:
        h = Hh.InProcEP(W, E, T, K, H, compact=mode == "compact", transport="push" if mode == "push" else "pull")
        got = h.dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).cuda() for i in idxs], Hh.QUANT_INT8)
        want = O.normal_dispatch(xs, idxs, E, True)
        rows = lambda g, w: (g["recv_x"], g["recv_x_scales"], g["recv_src_idx"], w.recv_x, w.recv_x_scales, w.total_recv)
    else:
        if mode != "replicated":
            pytest.skip("one low-latency form")
        h = Hh.InProcEP(W, E, T, K, H)
        got = h.ll_dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).int().cuda() for i in idxs], Hh.QUANT_INT8_NOEPS, 1)
        want = O.low_latency_dispatch(xs, idxs, T, E, True)
        rows = lambda g, w: (g["packed_recv_x"], g["packed_recv_x_scales"], g["src_info"], w.packed_recv_x, w.packed_recv_x_scales, w.total)
    seen = 0
    for r in range(W):
        gx, gs, gi, wx, wsc, n = rows(got[r], want[r])
        gx, gs, tri = gx.cpu().numpy()[:n], gs.cpu().numpy()[:n], gi.cpu().numpy()[:3 * n].reshape(-1, 3)
        assert np.array_equal(gs.view(np.uint32), wsc[:n].view(np.uint32))
        mask = np.ones((n, H), bool)
        for j in range(n):
            src, t = int(tri[j, 0]), int(tri[j, 1])
            if t in bad[src]:
                mask[j, bad[src][t]] = False
                assert np.isinf(gs[j]) and gs[j] > 0 and not gx[j][mask[j]].any()
                seen += 1
        assert np.array_equal(gx[mask], wx[:n][mask])
    assert seen == 4 * K * W
