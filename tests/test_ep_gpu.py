"""GPU parity tests: HIP dispatch/combine kernels (through the C-ABI) vs the CPU oracle, bit-exact.
W ranks are simulated in one process on one GPU (tests/ep_harness.py)."""
import numpy as np
import pytest
import torch

from oracle import ep as O
from oracle.bf16 import bf16_bits_to_f32, f32_to_bf16_bits_rne, torch_to_bits, bits_to_torch

pytestmark = pytest.mark.gpu


def make_topk(rng, T, K, E, drop=0.0, active=None):
    scores = np.abs(rng.standard_normal((T, E))) + 1
    if active is not None:
        mask = np.zeros(E, bool)
        mask[active] = True
        scores[:, ~mask] = 0
    idx = np.argsort(-scores, axis=1, kind="stable")[:, :K].astype(np.int64)
    if drop > 0:
        idx[rng.random((T, K)) < drop] = -1
    return idx


def rand_bits(rng, shape, scale=1.0):
    return f32_to_bf16_bits_rne((rng.standard_normal(shape) * scale).astype(np.float32))


def dev_bf16(bits):
    return bits_to_torch(bits).cuda()


LAYOUT_CASES = [(0, 4, 16, 4, 0), (1, 1, 2, 2, 0), (4, 2, 8, 2, 0), (33, 8, 64, 8, 0.3), (256, 2, 8, 1, 0),
                (257, 8, 256, 8, 0.1), (4096, 8, 256, 8, 0.0), (1000, 16, 1024, 16, 0.2), (8192, 8, 256, 8, 0.05),
                (129, 3, 24, 4, 0.5),
                # the cooperative single launch at its edges: one token past a workgroup (1025), a ragged last workgroup, 1024 experts
                # (the LDS limit of the 16-unit workgroup), 2048 experts (falls back to three launches), its 128-workgroup ceiling
                (1025, 8, 256, 8, 0.1), (5000, 6, 384, 8, 0.2), (3000, 16, 1024, 64, 0.0), (2500, 8, 2048, 8, 0.1),
                (131072, 2, 64, 8, 0.3)]


@pytest.mark.parametrize("T,K,E,W,drop", LAYOUT_CASES)
@pytest.mark.parametrize("i32", [False, True])
@pytest.mark.parametrize("coop", [False, True], ids=["three_launches", "one_launch"])
def test_dispatch_layout_bit_exact(T, K, E, W, drop, i32, coop):
    import ep_harness as Hh
    rng = np.random.default_rng(T * 31 + K + E)
    idx = make_topk(rng, T, K, E, drop) if T else np.zeros((0, K), np.int64)
    if T > 5:
        idx[3, :] = -1                     # a token selecting nothing
        idx[4, 0] = E + 5                  # out-of-range id is skipped like -1 (dispatch_layout.h:166)
    want = O.dispatch_layout(idx, E, W)
    t = torch.from_numpy(idx).cuda()
    if i32:
        t = t.int()
    for rep in range(2):                   # twice on the same sync words: the grid barrier re-armed itself
        got = Hh.layout(t, E, W, coop=coop)
        torch.cuda.synchronize()
        for k in ("num_tokens_per_rank", "num_tokens_per_expert", "is_token_in_rank", "send_token_idx_small"):
            assert np.array_equal(got[k].cpu().numpy(), want[k]), (k, rep)
        assert np.array_equal(got["send_data_offset"].cpu().numpy(), O.send_data_offset(want["num_tokens_per_expert"]))
    assert not Hh._SYNC["words"].any()


def test_dispatch_layout_barrier_timeout_is_reported():
    """A grid barrier that cannot close properly (here: an arrival word that is not zero when lent -- the first workgroups see more
    arrivals than the grid has workgroups, the last one sees the wrapped count and runs into the 2 s bound) must not return plausible
    tables silently: the kernel stores MI_EP_STATUS_LAYOUT_BARRIER in the caller's status word and sets
    the count tables to -1; the pair of words re-arms itself, and the next launch that borrows it is correct again."""
    import ep_harness as Hh
    T, K, E, W = 4096, 8, 256, 8
    rng = np.random.default_rng(5)
    idx = make_topk(rng, T, K, E, 0.0)
    t = torch.from_numpy(idx).cuda()
    B = (T + 16 * 16 - 1) // (16 * 16)                     # workgroups of 16 units of 16 tokens
    words = torch.tensor([-B, 0], dtype=torch.int32, device="cuda")
    status = torch.zeros(4, dtype=torch.int32, device="cuda")
    got = Hh.layout(t, E, W, coop=True, status=status, words=words)
    torch.cuda.synchronize()
    assert int(status[0]) == 6000                           # MI_EP_STATUS_LAYOUT_BARRIER
    assert (got["num_tokens_per_expert"] == -1).all() and (got["num_tokens_per_rank"] == -1).all()
    assert not words.any()                                  # re-armed by the last workgroup to leave
    status.zero_()
    got = Hh.layout(t, E, W, coop=True, status=status, words=words)
    torch.cuda.synchronize()
    want = O.dispatch_layout(idx, E, W)
    assert int(status[0]) == 0 and np.array_equal(got["num_tokens_per_expert"].cpu().numpy(), want["num_tokens_per_expert"])
    assert np.array_equal(got["send_token_idx_small"].cpu().numpy(), want["send_token_idx_small"])


DISPATCH_CASES = [
    # W, T, H, K, E, drop, active
    (1, 256, 1024, 2, 8, 0.0, None),        # BASELINE C1 shape
    (2, 33, 128, 2, 4, 0.0, None),
    (4, 64, 256, 8, 32, 0.2, None),
    (8, 40, 7168, 8, 256, 0.0, None),       # C2 row format (H=7168, E=256), few tokens
    (8, 17, 128, 4, 8, 0.3, None),          # L = 1
    (8, 64, 512, 8, 64, 0.0, [0, 1, 2, 3, 8, 9]),   # skewed routing (--active-ranks style)
    (3, 50, 2048, 6, 12, 0.1, None),        # non power-of-two world
    (2, 5, 8192, 16, 32, 0.0, None),        # max hidden, max top-k
    (8, 20, 128, 8, 2048, 0.1, None),       # the most experts: the count exchange keeps W * (E + 1) counts in LDS (84 KB, above the 64 KB default)
]


def _run_dispatch(W, T, H, K, E, drop, active, quant_mode, ragged=True, compact=False, transport="pull"):
    import ep_harness as Hh
    rng = np.random.default_rng(W * 1000 + T)
    Ts = [T + (r if ragged else 0) for r in range(W)]
    if ragged and W > 1:
        Ts[-1] = 0                                    # an empty rank
    xs = [rand_bits(rng, (t, H), 3.0) for t in Ts]
    for x in xs:
        if x.shape[0] > 2:
            x[1, :] = 0                               # an all-zero row (amax = 0)
    idxs = [make_topk(rng, t, K, E, drop, active) if t else np.zeros((0, K), np.int64) for t in Ts]
    ws = [rng.standard_normal((t, K)).astype(np.float32) for t in Ts]
    h = Hh.InProcEP(W, E, max(Ts) + 1, K, H, compact=compact, transport=transport)
    got = h.dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).cuda() for i in idxs], quant_mode)
    return h, xs, idxs, ws, got


@pytest.mark.parametrize("W,T,H,K,E,drop,active", DISPATCH_CASES)
@pytest.mark.parametrize("quant", [False, True])
@pytest.mark.parametrize("mode", ["replicated", "compact", "push"])
def test_normal_dispatch_combine_bit_exact(W, T, H, K, E, drop, active, quant, mode):
    """stage + pull (one staged row per (t, k): the all-to-all transport's format), stage_compact + pull_indexed (one row per
    token + index in the sender's window, pulled by the receivers) and stage_push + local pull_indexed (the same row + index
    written into the receivers' windows) must all reproduce the oracle bit for bit."""
    import ep_harness as Hh
    qm = Hh.QUANT_INT8 if quant else Hh.QUANT_NONE
    h, xs, idxs, ws, got = _run_dispatch(W, T, H, K, E, drop, active, qm, compact=mode == "compact",
                                         transport="push" if mode == "push" else "pull")
    want = O.normal_dispatch(xs, idxs, E, quant)
    for r in range(W):
        g, w = got[r], want[r]
        assert g["total"] == w.total_recv
        for k in ("recv_count", "recv_offset", "recv_tokens_per_expert", "expert_global_offset",
                  "srcrank_in_expert_offset", "r_in_srcrank_offset", "total_recv_token", "max_bs"):
            assert np.array_equal(g["tables"][k].cpu().numpy().reshape(-1), np.asarray(w.notify[k]).reshape(-1)), (r, k)
        if mode == "push":
            # relative pull offsets: position of segment (le, src) among the rows `src` sends to this rank
            L_ = E // W
            cnt_m = g["cnt"].cpu().numpy()
            rel = np.zeros(L_ * W, np.int32)
            for src in range(W):
                rel[src::W] = np.concatenate([[0], np.cumsum(cnt_m[src, r * L_:(r + 1) * L_])[:-1]])
            assert np.array_equal(g["tables"]["pull_offset"].cpu().numpy(), rel)
        n = w.total_recv
        assert np.array_equal(g["recv_src_idx"].cpu().numpy()[:3 * n], w.recv_src_idx[:3 * n])
        if quant:
            assert np.array_equal(g["recv_x"].cpu().numpy()[:n], w.recv_x[:n])
            assert np.array_equal(g["recv_x_scales"].cpu().numpy()[:n].view(np.uint32), w.recv_x_scales[:n].view(np.uint32))
        else:
            assert np.array_equal(torch_to_bits(g["recv_x"])[:n], w.recv_x[:n])
    # expert side: de-quantise (reference test convention) and combine
    if quant:
        ys_np = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) for w in want]
    else:
        ys_np = [w.recv_x for w in want]
    comb_want = O.combine(ys_np, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
    comb_got = h.combine([dev_bf16(y) for y in ys_np], [g["recv_src_idx"] for g in got], [g["total"] for g in got],
                         [torch.from_numpy(i).cuda() for i in idxs], [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), r
        if xs[r].shape[0]:
            d = O.calc_diff(bf16_bits_to_f32(comb_want[r]), O.golden_combined(xs[r], idxs[r], ws[r]))
            assert d < (3e-3 if quant else 1e-5)


def test_combine_without_weights_uses_ones():
    import ep_harness as Hh
    W, T, H, K, E = 2, 16, 128, 4, 8
    h, xs, idxs, ws, got = _run_dispatch(W, T, H, K, E, 0.2, None, Hh.QUANT_NONE, ragged=False)
    want = O.normal_dispatch(xs, idxs, E, False)
    comb_want = O.combine([w.recv_x for w in want], [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs,
                          [None] * W, E)
    comb_got = h.combine([g["recv_x"] for g in got], [g["recv_src_idx"] for g in got], [g["total"] for g in got],
                         [torch.from_numpy(i).cuda() for i in idxs], [None] * W)
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r])


LL_CASES = [(2, 16, 128, 2, 8, 0.0), (8, 128, 7168, 8, 256, 0.0), (8, 128, 512, 8, 64, 0.3), (4, 1, 256, 4, 16, 0.0),
            (8, 2, 1024, 8, 8, 0.0),
            # edges of the one-launch send: one selection per token, one local expert per rank, a rank without tokens (T = 3: rank 1 has none),
            # the largest batch it takes (1024 tokens), more experts than a 1024-thread workgroup has lanes
            (2, 33, 256, 1, 8, 0.0), (4, 20, 512, 4, 4, 0.2), (2, 3, 128, 2, 4, 0.0), (2, 1024, 128, 8, 64, 0.1), (2, 64, 256, 8, 1024, 0.0)]


@pytest.mark.parametrize("W,T,H,K,E,drop", LL_CASES)
@pytest.mark.parametrize("quant", [False, True])
@pytest.mark.parametrize("count_type,fused", [(1, True), (1, False), (0, True)])
def test_low_latency_dispatch_combine_bit_exact(W, T, H, K, E, drop, quant, count_type, fused):
    import ep_harness as Hh
    rng = np.random.default_rng(W * 77 + T)
    Ts = [T] * W
    if T > 1:
        Ts[0] = T - 1                                   # fewer tokens than num_max_dispatch_tokens_per_rank
    if T == 3:
        Ts[1] = 0                                       # a rank that sends nothing
    xs = [rand_bits(rng, (t, H), 2.0) for t in Ts]
    idxs = [make_topk(rng, t, K, E, drop) for t in Ts]
    ws = [np.abs(rng.standard_normal((t, K))).astype(np.float32) for t in Ts]
    h = Hh.InProcEP(W, E, T, K, H)
    qm = Hh.QUANT_INT8_NOEPS if quant else Hh.QUANT_NONE
    # the reference casts topk_idx to int32 for LL (low_latency_strategy.py:57)
    got = h.ll_dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).int().cuda() for i in idxs], qm, count_type, fused=fused)
    want = O.low_latency_dispatch(xs, idxs, T, E, quant, expert_token_nums_type=count_type)
    for r in range(W):
        g, w = got[r], want[r]
        n = w.total
        assert np.array_equal(g["layout_range"].cpu().numpy(), w.layout_range)
        assert np.array_equal(g["packed_recv_count"].cpu().numpy(), w.packed_recv_count)
        assert np.array_equal(g["src_info"].cpu().numpy()[:3 * n], w.src_info)
        if quant:
            assert np.array_equal(g["packed_recv_x"].cpu().numpy()[:n], w.packed_recv_x[:n])
            assert np.array_equal(g["packed_recv_x_scales"].cpu().numpy()[:n].view(np.uint32),
                                  w.packed_recv_x_scales[:n].view(np.uint32))
        else:
            assert np.array_equal(torch_to_bits(g["packed_recv_x"])[:n], w.packed_recv_x[:n])
    ys_np = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) if quant else w.packed_recv_x for w in want]
    comb_want = O.combine(ys_np, [w.src_info for w in want], [w.total for w in want], idxs, ws, E)
    comb_got = h.combine([dev_bf16(y) for y in ys_np], [g["src_info"] for g in got], [w.total for w in want],
                         [torch.from_numpy(i).int().cuda() for i in idxs], [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r])


@pytest.mark.parametrize("quant", [True, False])
@pytest.mark.parametrize("compact", ["replicated", "compact", "push"])
def test_full_size_c2_properties(quant, compact):
    """BASELINE C2 sizes (W=8, 4096 tok/rank, H=7168, top-8, E=256), for both staging formats (compact = stage_compact +
    pull_indexed, the path the host runtime and bench.py use): size-independent properties --
    counts == global histogram, ordering contract, int8 payload == quantised source row, round trip
    closed form (reference tests' golden) and run-to-run determinism."""
    import ep_harness as Hh
    W, T, H, K, E = 8, 4096, 7168, 8, 256
    L = E // W
    g = torch.Generator(device="cuda").manual_seed(1234)
    xs = [torch.randn((T, H), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16) for _ in range(W)]
    idxs = [torch.topk(torch.randn((T, E), generator=g, device="cuda").abs() + 1, K, dim=-1, sorted=False)[1] for _ in range(W)]
    ws = [torch.randn((T, K), generator=g, device="cuda") for _ in range(W)]
    h = Hh.InProcEP(W, E, T, K, H, compact=compact == "compact", transport="push" if compact == "push" else "pull")
    qm = Hh.QUANT_INT8 if quant else Hh.QUANT_NONE
    got = h.dispatch(xs, idxs, qm)
    hist = sum(torch.bincount(i.reshape(-1), minlength=E) for i in idxs)
    for r in range(W):
        per_e = got[r]["tables"]["recv_tokens_per_expert"].long()
        assert torch.equal(per_e, hist[r * L:(r + 1) * L])
    for r in range(W):
        n = got[r]["total"]
        tri = got[r]["recv_src_idx"][:3 * n].view(-1, 3).long()
        e_row = torch.empty(n, dtype=torch.long, device="cuda")
        src_x = torch.empty((n, H), dtype=torch.bfloat16, device="cuda")
        for s in range(W):
            m = tri[:, 0] == s
            e_row[m] = idxs[s][tri[m, 1], tri[m, 2]]
            src_x[m] = xs[s][tri[m, 1]]
        key = ((e_row - r * L) * W + tri[:, 0]) * (T * K) + tri[:, 1] * K + tri[:, 2]
        assert bool((key[1:] > key[:-1]).all())           # (local expert, src rank, row-major (t,k)) order
        if quant:
            # payload == oracle quantisation of the source row (sampled: oracle on 2048 rows)
            sel = torch.randperm(n, device="cuda")[:2048]
            q_want, s_want = O.quant_int8_rows(torch_to_bits(src_x[sel]), 1e-12)
            assert np.array_equal(got[r]["recv_x"][sel].cpu().numpy(), q_want)
            assert np.array_equal(got[r]["recv_x_scales"][sel].cpu().numpy().view(np.uint32), s_want.view(np.uint32))
        else:
            assert torch.equal(got[r]["recv_x"][:n], src_x)
    # round trip
    if quant:
        ys = [(g_["recv_x"].float() * g_["recv_x_scales"][:, None]).to(torch.bfloat16) for g_ in got]
    else:
        ys = [g_["recv_x"] for g_ in got]
    comb = h.combine(ys, [g_["recv_src_idx"] for g_ in got], [g_["total"] for g_ in got], idxs, ws)
    for r in range(W):
        golden = xs[r].float() * ws[r].sum(dim=1, keepdim=True)
        a, b = comb[r].double() + 1, golden.double() + 1
        diff = 1 - 2 * (a * b).sum() / (a * a + b * b).sum()
        assert diff.item() < (3e-3 if quant else 1e-5)
    # determinism: a second dispatch gives identical bytes
    got2 = h.dispatch(xs, idxs, qm)
    for r in range(W):
        assert torch.equal(got[r]["recv_x"], got2[r]["recv_x"]) and torch.equal(got[r]["recv_src_idx"], got2[r]["recv_src_idx"])


@pytest.mark.parametrize("W,T,H,K,E,drop,active", DISPATCH_CASES[:7])
@pytest.mark.parametrize("quant", [False, True])
def test_alltoall_transport_kernels_bit_exact(W, T, H, K, E, drop, active, quant):
    """The RCCL-fallback pipeline (per-source staging + relative pull offsets, combine_pack, gather-mode reduce) at W > 1."""
    import ep_harness as Hh
    rng = np.random.default_rng(W * 31 + T)
    Ts = [T + r for r in range(W)]
    xs = [rand_bits(rng, (t, H), 3.0) for t in Ts]
    idxs = [make_topk(rng, t, K, E, drop, active) for t in Ts]
    ws = [rng.standard_normal((t, K)).astype(np.float32) for t in Ts]
    a2a = Hh.InProcA2A(W, E, K, H)
    qm = Hh.QUANT_INT8 if quant else Hh.QUANT_NONE
    got = a2a.dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).cuda() for i in idxs], qm)
    want = O.normal_dispatch(xs, idxs, E, quant)
    for r in range(W):
        n = want[r].total_recv
        assert got[r]["total"] == n
        assert np.array_equal(got[r]["send_head"].cpu().numpy()[:E], want[r].send_head)
        assert np.array_equal(got[r]["recv_src_idx"].cpu().numpy()[:3 * n], want[r].recv_src_idx[:3 * n])
        if quant:
            assert np.array_equal(got[r]["recv_x"].cpu().numpy()[:n], want[r].recv_x[:n])
            assert np.array_equal(got[r]["recv_x_scales"].cpu().numpy()[:n].view(np.uint32), want[r].recv_x_scales[:n].view(np.uint32))
        else:
            assert np.array_equal(torch_to_bits(got[r]["recv_x"])[:n], want[r].recv_x[:n])
    ys_np = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
    comb_want = O.combine(ys_np, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
    comb_got = a2a.combine([dev_bf16(y) for y in ys_np], got, [torch.from_numpy(i).cuda() for i in idxs],
                           [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), r


# ---- committed fixtures: HIP kernels (through the C-ABI) vs tests/golden/ep_case_*.npz ------------------------------------
import glob as _glob
import os as _os

_GOLD = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("path", sorted(_glob.glob(_os.path.join(_GOLD, "ep_case_*.npz"))))
def test_kernels_reproduce_committed_fixtures(path):
    import ep_harness as Hh
    z = np.load(path)
    W, H, K, E, quant, MT = int(z["W"]), int(z["H"]), int(z["K"]), int(z["E"]), bool(z["quant"]), int(z["max_tokens"])
    xs, idxs, ws = [z[f"x{r}"] for r in range(W)], [z[f"idx{r}"] for r in range(W)], [z[f"w{r}"] for r in range(W)]
    h = Hh.InProcEP(W, E, MT, K, H)
    got = h.dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).cuda() for i in idxs], Hh.QUANT_INT8 if quant else Hh.QUANT_NONE)
    ys = []
    for r in range(W):
        n = len(z[f"recv_src_idx{r}"]) // 3
        assert got[r]["total"] == n
        assert np.array_equal(got[r]["recv_src_idx"].cpu().numpy()[:3 * n], z[f"recv_src_idx{r}"])
        assert np.array_equal(got[r]["tables"]["recv_count"].cpu().numpy().reshape(-1), z[f"send_head{r}"])
        if quant:
            assert np.array_equal(got[r]["recv_x"].cpu().numpy()[:n], z[f"recv_x{r}"][:n])
            assert np.array_equal(got[r]["recv_x_scales"].cpu().numpy()[:n].view(np.uint32), z[f"recv_scales{r}"][:n].view(np.uint32))
            ys.append(O.per_token_cast_back(z[f"recv_x{r}"], z[f"recv_scales{r}"]))
        else:
            assert np.array_equal(torch_to_bits(got[r]["recv_x"])[:n], z[f"recv_x{r}"][:n])
            ys.append(z[f"recv_x{r}"])
    comb = h.combine([dev_bf16(y) for y in ys], [g["recv_src_idx"] for g in got], [g["total"] for g in got],
                     [torch.from_numpy(i).cuda() for i in idxs], [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb[r]), z[f"combined{r}"]), r
    qm = Hh.QUANT_INT8_NOEPS if quant else Hh.QUANT_NONE
    ll = h.ll_dispatch([dev_bf16(x) for x in xs], [torch.from_numpy(i).int().cuda() for i in idxs], qm, 1)
    for r in range(W):
        assert np.array_equal(ll[r]["layout_range"].cpu().numpy(), z[f"ll_layout_range{r}"])
        assert np.array_equal(ll[r]["packed_recv_count"].cpu().numpy(), z[f"ll_recv_count{r}"])
        n = len(z[f"ll_src_info{r}"]) // 3
        assert np.array_equal(ll[r]["src_info"].cpu().numpy()[:3 * n], z[f"ll_src_info{r}"])


@pytest.mark.parametrize("W,T,H,K,E,drop", LL_CASES)
@pytest.mark.parametrize("quant", [False, True])
def test_low_latency_two_launch_forms_bit_exact(W, T, H, K, E, drop, quant):
    """mi_ep_ll_dispatch_layout_send_tagged + mi_ep_ll_wait_pack and mi_ep_combine_push_flagged + mi_ep_combine_reduce_flagged through the
    C-ABI: same tables, rows and sums as the oracle, three calls in a row on one set of windows (device-resident call counters, both ping-pong
    halves, tags and flag words of earlier calls left in place)."""
    import ep_harness as Hh
    rng = np.random.default_rng(W * 79 + T)
    h = Hh.InProcEP(W, E, T, K, H)
    qm = Hh.QUANT_INT8_NOEPS if quant else Hh.QUANT_NONE
    for call in range(3):
        Ts = [T] * W
        if T > 1:
            Ts[0] = T - 1
        if T == 3 or call == 1:
            Ts[-1] = 0                                  # a rank that sends nothing
        xs = [rand_bits(rng, (t, H), 2.0) for t in Ts]
        idxs = [make_topk(rng, t, K, E, drop) for t in Ts]
        ws = [np.abs(rng.standard_normal((t, K))).astype(np.float32) for t in Ts]
        dev_idx = [torch.from_numpy(i).int().cuda() for i in idxs]
        got = h.ll_dispatch_tagged([dev_bf16(x) for x in xs], dev_idx, qm, 1)
        want = O.low_latency_dispatch(xs, idxs, T, E, quant, expert_token_nums_type=1)
        for r in range(W):
            g, w = got[r], want[r]
            n = w.total
            assert np.array_equal(g["layout_range"].cpu().numpy(), w.layout_range), (call, r)
            assert np.array_equal(g["packed_recv_count"].cpu().numpy(), w.packed_recv_count), (call, r)
            assert np.array_equal(g["src_info"].cpu().numpy()[:3 * n], w.src_info), (call, r)
            if quant:
                assert np.array_equal(g["packed_recv_x"].cpu().numpy()[:n], w.packed_recv_x[:n]), (call, r)
                assert np.array_equal(g["packed_recv_x_scales"].cpu().numpy()[:n].view(np.uint32), w.packed_recv_x_scales[:n].view(np.uint32))
            else:
                assert np.array_equal(torch_to_bits(g["packed_recv_x"])[:n], w.packed_recv_x[:n]), (call, r)
        ys_np = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) if quant else w.packed_recv_x for w in want]
        comb_want = O.combine(ys_np, [w.src_info for w in want], [w.total for w in want], idxs, ws, E)
        comb_got = h.combine_flagged([dev_bf16(y) for y in ys_np], [g["src_info"] for g in got], [w.total for w in want], dev_idx,
                                     [torch.from_numpy(w_).cuda() for w_ in ws])
        for r in range(W):
            assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), (call, r)


# ---- start-up self-test of mapped windows (C-ABI): W simulated ranks, one stream each (a rank's check kernel waits for its peers') ------
def _selftest_run(W, rounds, first_epoch, state, bad_rank=None, stale_rank=None):
    import ep_harness as Hh
    from ctypes import c_int, c_size_t, c_uint32, c_uint64, c_void_p
    from capi import ptr, ptr_array
    L_ = Hh.lib()
    L_.mi_ep_selftest_bytes.restype = c_size_t
    L_.mi_ep_selftest_bytes.argtypes = [c_int]
    L_.mi_ep_selftest.restype = c_int
    L_.mi_ep_selftest.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_uint64, c_int, c_uint32, c_void_p, c_int, c_void_p]
    if not state:
        nb = L_.mi_ep_selftest_bytes(W)
        state.update(rows=[torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(W)],
                     flags=[torch.zeros(64, dtype=torch.int64, device="cuda") for _ in range(W)],
                     acks=[torch.zeros(64, dtype=torch.int64, device="cuda") for _ in range(W)],
                     # one stream per simulated rank, on DIFFERENT hardware queues: the runtime deals plain streams round-robin onto four
                     # queues, so which two a test gets depends on how many streams the process made before it; streams of different
                     # priorities never share a queue
                     streams=[torch.cuda.Stream(priority=(0 if r % 2 == 0 else -1)) for r in range(W)])
    status = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(W)]
    torch.cuda.synchronize()
    rows_p, flags_p, acks_p = ptr_array([t.data_ptr() for t in state["rows"]]), ptr_array([t.data_ptr() for t in state["flags"]]), \
        ptr_array([t.data_ptr() for t in state["acks"]])
    for r in range(W):
        tag = 0x5E1F0000 + first_epoch + (77 if r == bad_rank else 0)
        rc = L_.mi_ep_selftest(rows_p, flags_p, ptr(state["flags"][r]), acks_p, ptr(state["acks"][r]), W, r, first_epoch, rounds, tag,
                               ptr(status[r]), 2000, c_void_p(state["streams"][r].cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    return [int(s[0]) for s in status]


def _selftest_inlaunch_run(W, rounds, first_epoch, state, stale_rank=None, stale_from=1):
    """mi_ep_selftest_inlaunch on the state of _selftest_run (same ack words, epochs continue): W simulated ranks, one stream each."""
    import ep_harness as Hh
    from ctypes import c_int, c_size_t, c_uint32, c_uint64, c_void_p
    from capi import ptr, ptr_array
    L_ = Hh.lib()
    for fn in (L_.mi_ep_selftest_inlaunch_bytes, L_.mi_ep_selftest_inlaunch_flag_words):
        fn.restype, fn.argtypes = c_size_t, [c_int]
    L_.mi_ep_selftest_inlaunch.restype = c_int
    L_.mi_ep_selftest_inlaunch.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_int, c_uint64, c_int, c_uint32,
                                           c_int, c_void_p, c_int, c_void_p]
    nb, nw = L_.mi_ep_selftest_inlaunch_bytes(W), L_.mi_ep_selftest_inlaunch_flag_words(W)
    if "il_rows" not in state:
        state.update(il_rows=[torch.zeros(2 * nb, dtype=torch.uint8, device="cuda") for _ in range(W)],
                     il_flags=[torch.zeros(2 * nw, dtype=torch.int32, device="cuda") for _ in range(W)])
    status = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(W)]
    torch.cuda.synchronize()
    rows_p, flags_p, acks_p = ptr_array([t.data_ptr() for t in state["il_rows"]]), ptr_array([t.data_ptr() for t in state["il_flags"]]), \
        ptr_array([t.data_ptr() for t in state["acks"]])
    for r in range(W):
        rc = L_.mi_ep_selftest_inlaunch(rows_p, nb, flags_p, nw * 4, acks_p, ptr(state["acks"][r]), W, r, first_epoch, rounds,
                                        0x1A7C0000 + first_epoch, stale_from if r == stale_rank else -1, ptr(status[r]), 2000,
                                        c_void_p(state["streams"][r].cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    return [int(s[0]) for s in status]


@pytest.mark.parametrize("W", [2])
def test_window_selftest_in_launch_handoff_leg(W):
    """mi_ep_selftest_inlaunch: tag / flag word behind a drained write-through payload, polled and read with system-scope loads inside ONE launch
    (the hand-off of the two-launch low-latency forms), four rounds over both ping-pong halves on the same addresses; a second call continues;
    a rank that raises its words WITHOUT rewriting the payload -- a stale line, as its consumers see it -- is reported as 7000 + s (tagged
    row) or 7500 + s (flagged row) by every rank that consumes from it (itself included) and nobody reports the healthy rank; rounds before the
    injection pass."""
    state = {}
    assert _selftest_run(W, 2, 1, state) == [0] * W                    # (first leg: leaves the ack words at epoch 2)
    assert _selftest_inlaunch_run(W, 4, 3, state) == [0] * W
    assert _selftest_inlaunch_run(W, 4, 7, state) == [0] * W
    codes = _selftest_inlaunch_run(W, 4, 11, state, stale_rank=1, stale_from=2)
    assert codes[0] in (7001, 7501), codes
    assert codes[1] in (7001, 7501), codes      # (a rank also produces for and consumes from itself: rank 1 sees its own stale rows; nobody names rank 0)
    assert _selftest_inlaunch_run(W, 2, 15, state) == [0] * W          # and the scratch is usable again afterwards


@pytest.mark.parametrize("W", [2])      # in ONE process two simulated ranks get their own hardware queues; more would share one (the runtime maps streams
def test_window_selftest_rounds_and_failure_codes(W):      # onto 4 queues) and a spinning check kernel would block its peer's post behind it.
                                                           # W = 4, 8: every multi-process deep_ep.Buffer test runs the self-test at start-up.
    """mi_ep_selftest: two rounds on the same addresses pass; a second call continues the epochs; a rank that writes a different
    pattern is reported by every other rank as a corrupt row (3000 + s) and sees corrupt read-backs itself; a rank that never
    shows up is a bounded timeout (1 + s), not a hang."""
    state = {}
    assert _selftest_run(W, 2, 1, state) == [0] * W
    assert _selftest_run(W, 2, 3, state) == [0] * W
    codes = _selftest_run(W, 1, 5, state, bad_rank=1)
    assert all(c != 0 for c in codes)
    assert all(c in (3001, 5001) for i, c in enumerate(codes) if i != 1), codes
    assert codes[1] // 1000 in (3, 4, 5), codes


# ---- quant_mode "pertoken_fp8_e4m3" (MI_EP_QUANT_FP8_E4M3): the reference's Ascend950-only per-token FP8 dispatch, served natively here ----
@pytest.mark.parametrize("W,T,H,K,E,drop,active", DISPATCH_CASES[:6])
@pytest.mark.parametrize("mode", ["replicated", "compact", "push", "low_latency"])
def test_fp8_e4m3_dispatch_bit_exact(W, T, H, K, E, drop, active, mode):
    """Payload bytes (OCP E4M3 bit patterns from v_cvt_pk_fp8_f32) and scale words against oracle.ep.quant_fp8_e4m3_rows through all
    three normal-dispatch forms and the low-latency slabs; the routing tables must not depend on the payload type; combine of the
    de-quantised rows round-trips like INT8 does."""
    import ep_harness as Hh
    rng = np.random.default_rng(W * 131 + T + H)
    Ts = [T + r for r in range(W)]
    xs = [rand_bits(rng, (t, H), float(rng.choice([0.01, 1.0, 300.0]))) for t in Ts]
    for x in xs:
        if x.shape[0] > 2:
            x[1, :] = 0                                  # an all-zero token: scale 1, zeros
            x[2, 5] = 0x7F7F                             # the largest finite bf16 in one element: everything else flushes towards zero
    idxs = [make_topk(rng, t, K, E, drop, active) for t in Ts]
    ws = [np.abs(rng.standard_normal((t, K))).astype(np.float32) for t in Ts]
    dev_idx = [torch.from_numpy(i).int().cuda() if mode == "low_latency" else torch.from_numpy(i).cuda() for i in idxs]
    if mode == "low_latency":
        MT = max(Ts)
        h = Hh.InProcEP(W, E, MT, K, H)
        got = h.ll_dispatch([dev_bf16(x) for x in xs], dev_idx, Hh.QUANT_FP8_E4M3, 1)
        want = O.low_latency_dispatch(xs, idxs, MT, E, "fp8")
        view = lambda g, w: (g["packed_recv_x"], g["packed_recv_x_scales"], g["src_info"], w.packed_recv_x, w.packed_recv_x_scales, w.src_info, w.total)
    else:
        h = Hh.InProcEP(W, E, max(Ts) + 1, K, H, compact=mode == "compact", transport="push" if mode == "push" else "pull")
        got = h.dispatch([dev_bf16(x) for x in xs], dev_idx, Hh.QUANT_FP8_E4M3)
        want = O.normal_dispatch(xs, idxs, E, "fp8")
        view = lambda g, w: (g["recv_x"], g["recv_x_scales"], g["recv_src_idx"], w.recv_x, w.recv_x_scales, w.recv_src_idx, w.total_recv)
    ys, tris, totals = [], [], []
    for r in range(W):
        gx, gs, gi, wx, wsc, wi, n = view(got[r], want[r])
        assert np.array_equal(gi.cpu().numpy()[:3 * n], wi[:3 * n])
        assert np.array_equal(gx.cpu().numpy().view(np.uint8)[:n], wx[:n]), (mode, r)
        assert np.array_equal(gs.cpu().numpy()[:n].view(np.uint32), wsc[:n].view(np.uint32))
        ys.append(O.per_token_cast_back(wx, wsc)), tris.append(gi), totals.append(n)
    comb_want = O.combine(ys, [view(got[r], want[r])[5] for r in range(W)], totals, idxs, ws, E)
    comb_got = h.combine([dev_bf16(y) for y in ys], tris, totals, dev_idx, [torch.from_numpy(w_).cuda() for w_ in ws])
    for r in range(W):
        assert np.array_equal(torch_to_bits(comb_got[r]), comb_want[r]), r


@pytest.mark.parametrize("T,K,E,W,S,rank,i32,with_w", [(33, 4, 12, 4, 1, 2, False, True), (1, 8, 48, 8, 2, 5, True, False), (700, 2, 4, 2, 1, 0, False, True),
                                                      (64, 7, 16, 8, 4, 7, True, True)])
def test_shared_expert_map_renames_experts(T, K, E, W, S, rank, i32, with_w):
    """mi_ep_shared_expert_map (MOE_SHARED_EXPERT_RANK_NUM; reference deep_ep.cpp:866-874, moe_distribute_dispatch_v2.h:555-604,650-696):
    routed expert e -> slot S*L + e (rank S + e / L), the shared selection (rank mod S) * L for tokens with an active selection, -1
    otherwise; weights copied with 1.0 appended (NULL in: ones); bad arguments are refused."""
    import ep_harness as Hh
    from capi import ptr, stream_ptr
    lib, ck = Hh.lib, Hh.ck
    L_ = E // (W - S)
    rng = np.random.default_rng(T + K)
    idx = rng.integers(-2, E + 2, (T, K))
    if T > 2:
        idx[1, :] = -1
    w = rng.standard_normal((T, K)).astype(np.float32)
    ti = torch.from_numpy(idx.astype(np.int32 if i32 else np.int64)).cuda()
    tw = torch.from_numpy(w).cuda()
    out = torch.full((T, K + 1), 7, dtype=torch.int32, device="cuda")
    wout = torch.zeros((T, K + 1), dtype=torch.float32, device="cuda")
    L = lib()
    ck(L.mi_ep_shared_expert_map(ptr(ti), int(i32), ptr(tw) if with_w else None, T, K, E, W, S, rank, ptr(out), ptr(wout), stream_ptr()))
    torch.cuda.synchronize()
    ok = (idx >= 0) & (idx < E)
    want = np.concatenate([np.where(ok, idx + S * L_, -1), np.where(ok.any(axis=1), (rank % S) * L_, -1)[:, None]], axis=1)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(wout.cpu().numpy(), np.concatenate([w if with_w else np.ones_like(w), np.ones((T, 1), np.float32)], axis=1))
    # every renamed id lands on the rank the reference sends it to
    dst = want[:, :K][ok] // L_
    assert np.array_equal(dst, idx[ok] // L_ + S)
    for bad in ((T, K, E, W, 0), (T, K, E, W, W), (T, K, E * (W - S) + 1, W, S) if W - S > 1 else (T, K, 0, W, S), (T, 16, E, W, S)):
        assert L.mi_ep_shared_expert_map(ptr(ti), int(i32), None, bad[0], bad[1], bad[2], bad[3], bad[4], 0, ptr(out), None, stream_ptr()) == -1
