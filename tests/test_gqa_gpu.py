"""GPU parity: HIP paged GQA decode (C-ABI mi_gqa_decode and the torch.ops.npu.decode_gqa / sgl_kernel_npu wrappers) vs
the CPU oracle, the committed outputs of the reference Triton kernel, and the reference test's golden at its tolerance
(rtol = atol = 1e-2, tests/python/sgl_kernel_npu/test_decode_attention.py:121)."""
import glob
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

import numpy as np
import pytest
import torch

from capi import load, ptr, stream_ptr
from oracle import kernels as OK

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load("libmi_sgl_kernels.so")
        _lib.mi_gqa_decode_workspace.restype = c_size_t
        _lib.mi_gqa_decode_workspace.argtypes = [c_int] * 4
        _lib.mi_gqa_decode_num_splits.argtypes = [c_int] * 4
        _lib.mi_gqa_decode.argtypes = [c_void_p] * 6 + [c_int] * 8 + [c_int64] * 10 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]
    return _lib


def run_gqa(q, k, v, lens, bt, sm_scale, num_splits=0):
    B, Hq, Lk = q.shape
    Hkv, Lv = k.shape[2], v.shape[3]
    out = torch.full((B, Hq, Lv), float("nan"), dtype=q.dtype, device=q.device)
    max_len = int(lens.max().item()) if B else 0
    L = lib()
    if num_splits == 0:
        num_splits = L.mi_gqa_decode_num_splits(B, Hq, Hkv, max_len)
    wsb = L.mi_gqa_decode_workspace(B, Hq, Lv, num_splits)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=q.device)
    rc = L.mi_gqa_decode(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lens), ptr(bt), B, Hq, Hkv, Lk, Lv, k.shape[1], bt.stride(0), max_len,
                         q.stride(0), q.stride(1), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
                         out.stride(0), out.stride(1), sm_scale, 0 if q.dtype == torch.bfloat16 else 1, num_splits, ptr(ws), wsb,
                         stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out


def exact_fp64(q, k, v, lens, bt, sm):
    B, Hq, _ = q.shape
    page, Hkv = k.shape[1], k.shape[2]
    group = Hq // Hkv
    out = torch.zeros((B, Hq, v.shape[3]), dtype=torch.float64)
    for b in range(B):
        L = int(lens[b])
        idx = bt[b, :(L + page - 1) // page].long()
        for kvh in range(Hkv):
            K = k[idx, :, kvh].reshape(-1, k.shape[3])[:L].double()
            V = v[idx, :, kvh].reshape(-1, v.shape[3])[:L].double()
            hs = slice(kvh * group, (kvh + 1) * group)
            out[b, hs] = torch.softmax((q[b, hs].double() @ K.T) * sm, -1) @ V
    return out


def check(got, want, exact, dtype):
    if dtype == torch.float16:
        assert torch.allclose(got.float(), want.float(), atol=1e-3, rtol=2 ** -10), (got.float() - want.float()).abs().max()
        return
    # bf16: same criterion as the MLA test -- P is rounded to 8 bits before P.V in the reference kernel, kernel and oracle
    # round against different (equally valid) running maxima, so both are measured against the exact fp64 result.
    err_k = (got.double() - exact).abs().max().item()
    err_o = (want.double() - exact).abs().max().item()
    assert err_k <= 1.5 * err_o + 1e-3, (err_k, err_o)
    assert torch.allclose(got.float(), want.float(), rtol=1e-2, atol=1e-2)


PLANNED = -1      # MI_MLA_SPLITS_PLANNED: the device-built, length-aware work list (decode_plan.h)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "gqa_ref_fp16_*.npz"))))
@pytest.mark.parametrize("splits", [1, 2, PLANNED])
def test_against_reference_kernel_outputs(path, splits):
    z = np.load(path)
    t = lambda key: torch.from_numpy(z[key]).cuda()
    k = t("k")
    v = k[..., :z["v"].shape[-1]] if int(z["v_is_view"]) else t("v")
    got = run_gqa(t("q"), k, v, t("kv_seq_lens"), t("block_table"), float(z["sm_scale"]), splits)
    want = torch.from_numpy(z["out"]).cuda()
    assert torch.allclose(got.float(), want.float(), atol=1e-3, rtol=2 ** -10), (got.float() - want.float()).abs().max()


CASES = [  # B, Hq, Hkv, Lk, Lv, S, page, ragged, v_is_view
    (2, 64, 8, 128, 128, 300, 128, True, True),       # reference config (16, 64, 8, 128, 128), v = k view
    (2, 128, 1, 288, 256, 260, 128, True, True),      # reference config (16, 128, 1, 288, 256): two head blocks per wave
    (2, 32, 1, 576, 512, 150, 64, True, False),       # 576/512 with an independent V cache -> generic kernel, 32-key tiles
    (3, 16, 2, 128, 128, 129, 16, True, False),
    (2, 8, 8, 64, 64, 70, 1, False, False),           # MHA, page_size 1
    (2, 24, 4, 80, 64, 200, 32, True, False),         # head dims padded inside the kernel (80 -> 128, 64 -> 128)
    (1, 40, 1, 192, 128, 333, 64, False, False),      # group of 40: partially filled head blocks
    (2, 16, 2, 256, 256, 1, 64, False, False),        # single key
    (1, 256, 1, 128, 128, 140, 64, False, False),     # group > 128: two workgroups per unit
    # large kv groups with dims in (192, 288] x <= 256 on power-of-two pages >= 32: the eight-wave kernel of gqa_decode_wide.hip
    (3, 128, 1, 288, 256, 700, 64, True, False),      # independent V cache, several tiles per split, ragged
    (2, 96, 1, 256, 256, 330, 32, True, False),       # 96 heads: two idle head waves; K rows narrower than the padded 288 (zeroed pad columns)
    (1, 256, 2, 288, 128, 200, 64, True, False),      # two kv heads of 128; V narrower than the padded 256
    (2, 200, 1, 224, 200, 97, 128, True, False),      # two head blocks (128 + 72 heads), odd dims (multiples of 8), S < page
    (2, 128, 1, 288, 256, 1, 64, False, True),        # a single key
    # V a column prefix of K (the reference test's own construction) on pages of >= 64 keys: the 64-key-tile instance (three slots, P.V from the K tile)
    (3, 128, 1, 288, 256, 700, 64, True, True),       # several tiles per split, ragged, the last tile partly filled
    (2, 96, 1, 256, 224, 330, 64, True, True),        # 96 heads (two idle head waves), dims below the padded (288, 256)
    (1, 256, 2, 288, 128, 450, 128, True, True),      # two kv heads of 128 q heads
    (2, 128, 1, 288, 256, 330, 32, True, True),       # pages of 32 keys: the view form on 32-key tiles
]


@pytest.mark.parametrize("B,Hq,Hkv,Lk,Lv,S,page,ragged,view", CASES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("splits", [0, 1, 3, PLANNED])
def test_against_oracle(B, Hq, Hkv, Lk, Lv, S, page, ragged, view, dtype, splits):
    torch.manual_seed(1)
    maxp = (S + page - 1) // page
    nb = B * maxp + 3
    q = torch.randn((B, Hq, Lk)).to(dtype)
    k = torch.randn((nb, page, Hkv, Lk)).to(dtype)
    v = k[..., :Lv] if view else torch.randn((nb, page, Hkv, Lv)).to(dtype)
    bt = torch.randperm(nb)[:B * maxp].to(torch.int32).reshape(B, maxp)
    lens = torch.tensor([max(1, S - 37 * i) if ragged else S for i in range(B)], dtype=torch.int32)
    sm = 1.0 / Lk ** 0.5
    want = OK.decode_gqa(q, k, v, lens, bt, sm)
    kc = k.cuda()
    vc = kc[..., :Lv] if view else v.cuda()
    got = run_gqa(q.cuda(), kc, vc, lens.cuda(), bt.cuda(), sm, splits).cpu()
    assert not torch.isnan(got.float()).any()
    check(got, want, exact_fp64(q, k, v, lens, bt, sm), dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D,Dv", [(32, 1, 576, 512), (128, 1, 576, 512), (128, 1, 288, 256), (64, 8, 128, 128)])
def test_python_entry_points_reference_configs(dtype, Hq, Hkv, D, Dv):
    """The reference's own test (test_decode_attention.py:63-128, configs :240-245, batch shrunk): decode_gqa and
    decode_gqa_high_performance vs decode_gqa_golden at rtol = atol = 1e-2.  (576, 512) with v = k[..., :512] takes the
    MLA kernel, the others the generic one."""
    from sgl_kernel_npu.attention.decode_attention import decode_gqa, decode_gqa_high_performance
    torch.manual_seed(1)
    B, S, page = 4, 1314, 128
    maxp = (S + page - 1) // page
    q = torch.randn((B, Hq, D), device="cuda").to(dtype)
    k = torch.randn((maxp * B, page, Hkv, D), device="cuda").to(dtype)
    v = k[..., :Dv]
    bt = torch.arange(B * maxp, dtype=torch.int32, device="cuda").reshape(B, maxp)
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    sm = 1.0 / D ** 0.5
    out = torch.empty((B, Hq, Dv), device="cuda", dtype=dtype)
    decode_gqa(q, k, v, out, lens, sm, page, bt)
    out1 = torch.empty_like(out)
    scratch = torch.empty((B, Hq, S), device="cuda", dtype=dtype)
    decode_gqa_high_performance(q, k, v, out1, lens, scratch, scratch, torch.empty_like(out), sm, page, bt)
    gold = OK.decode_gqa_golden(q.cpu(), k.cpu(), v.cpu(), lens.cpu(), bt.cpu(), sm)
    assert torch.allclose(out.cpu().float(), gold.float(), rtol=1e-2, atol=1e-2)
    assert torch.equal(out, out1)


def test_full_size_vs_fp32():
    """Llama-style decode at serving size: B=64, 64 q heads / 8 kv heads, D=128, 4096 keys, random page table."""
    B, Hq, Hkv, D, S, page = 64, 64, 8, 128, 4096, 64
    maxp = S // page
    nb = B * maxp
    g = torch.Generator(device="cuda").manual_seed(5)
    q = torch.randn((B, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
    bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(B, maxp)
    lens = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32)
    got = run_gqa(q, k, v, lens, bt, D ** -0.5)
    for b in (0, 31, 63):
        L = int(lens[b])
        idx = bt[b, :(L + page - 1) // page].long()
        for kvh in (0, 7):
            K = k[idx, :, kvh].reshape(-1, D)[:L].float()
            V = v[idx, :, kvh].reshape(-1, D)[:L].float()
            hs = slice(kvh * 8, kvh * 8 + 8)
            ref = torch.softmax((q[b, hs].float() @ K.T) * D ** -0.5, -1) @ V
            assert torch.allclose(got[b, hs].float(), ref, atol=1e-3, rtol=2 ** -7), (got[b, hs].float() - ref).abs().max()


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hkv,D,page,S,window,sink_dt", [
    (4, 64, 8, 64, 128, 700, -1, torch.bfloat16),          # GPT-OSS head shape (64 q / 8 kv heads of 64)
    (4, 64, 8, 64, 128, 700, 128, torch.float32),
    (3, 16, 2, 128, 16, 333, 100, torch.float32),           # window start inside a tile and inside a page
    (2, 8, 8, 64, 64, 40, 64, torch.float16),               # sequences shorter than the window
    (5, 32, 4, 128, 32, 2100, 1024, torch.bfloat16),        # several KV splits
])
def test_attention_sinks_decode(B, Hq, Hkv, D, page, S, window, sink_dt, dt):
    """attention/sinks_attention.py:90-137 through its Python entry point against the fp32 restatement: sink in the denominator, sliding window."""
    from sgl_kernel_npu.attention.sinks_attention import attention_sinks_triton
    torch.manual_seed(B * 7 + S)
    maxp = (S + page - 1) // page
    nb = B * maxp + 2
    q = torch.randn(B, Hq * D).to(dt)
    kc, vc = torch.randn(nb, page, Hkv, D).to(dt), torch.randn(nb, page, Hkv, D).to(dt)
    bt = torch.randperm(nb)[:B * maxp].reshape(B, maxp).to(torch.int32)
    lens = torch.tensor([max(1, S - 97 * i) for i in range(B)], dtype=torch.int32)
    sinks = (torch.randn(Hq) * 2).to(sink_dt)
    scale = D ** -0.5
    got = attention_sinks_triton(q.cuda(), kc.cuda(), vc.cuda(), sinks.cuda(), bt.cuda(), lens.cuda(), scale, window, Hq, Hkv)
    want = OK.attention_sinks(q, kc, vc, sinks, bt, lens, scale, window, Hq, Hkv)
    assert got.shape == (B, Hq * D) and got.dtype == dt
    tol = dict(atol=2e-3, rtol=2 ** -6 if dt == torch.bfloat16 else 2 ** -9)
    assert torch.allclose(got.cpu().float(), want.float(), **tol), (got.cpu().float() - want.float()).abs().max()


def test_attention_sinks_without_a_sink_equals_decode_gqa():
    """sinks = -inf, no window: the same kernel must give decode_gqa's bits."""
    from sgl_kernel_npu.attention.decode_attention import decode_gqa
    from sgl_kernel_npu.attention.sinks_attention import attention_sinks_triton
    torch.manual_seed(3)
    B, Hq, Hkv, D, page, S = 3, 32, 4, 128, 64, 900
    dt = torch.bfloat16
    maxp = (S + page - 1) // page
    nb = B * maxp
    q = torch.randn(B, Hq, D).to(dt).cuda()
    kc, vc = torch.randn(nb, page, Hkv, D).to(dt).cuda(), torch.randn(nb, page, Hkv, D).to(dt).cuda()
    bt = torch.randperm(nb)[:B * maxp].reshape(B, maxp).to(torch.int32).cuda()
    lens = torch.tensor([900, 1, 517], dtype=torch.int32).cuda()
    out = torch.empty(B, Hq, D, dtype=dt, device="cuda")
    decode_gqa(q, kc, vc, out, lens, D ** -0.5, page, bt)
    got = attention_sinks_triton(q.reshape(B, Hq * D), kc, vc, torch.full((Hq,), float("-inf"), device="cuda"), bt, lens, D ** -0.5, -1, Hq, Hkv)
    assert torch.equal(got.reshape(B, Hq, D), out)


@pytest.mark.parametrize("window", [-1, 48])
def test_attention_sinks_extend(window):
    """attention/sinks_attention.py:241-286: every new token of every sequence attends causally (its own length, its own window)."""
    from sgl_kernel_npu.attention.sinks_attention import attention_sinks_prefill_triton
    torch.manual_seed(11)
    Hq, Hkv, D, page = 16, 4, 64, 16
    dt = torch.bfloat16
    seq_lens = torch.tensor([5, 1, 37], dtype=torch.int32)          # new tokens
    ctx = torch.tensor([60, 1, 37], dtype=torch.int32)              # keys in the cache including them
    Bn, S = 3, int(seq_lens.sum())
    maxp = 8
    nb = Bn * maxp
    q = torch.randn(S, Hq * D).to(dt)
    kc, vc = torch.randn(nb, page, Hkv, D).to(dt), torch.randn(nb, page, Hkv, D).to(dt)
    bt = torch.randperm(nb).reshape(Bn, maxp).to(torch.int32)
    sinks = torch.randn(Hq)
    scale = D ** -0.5
    got = attention_sinks_prefill_triton(q.cuda(), kc.cuda(), vc.cuda(), sinks.cuda(), seq_lens.cuda(), bt.cuda(), ctx.cuda(), scale, window, Hq, Hkv)
    kv, rows = [], []
    for b in range(Bn):
        for t in range(int(seq_lens[b])):
            kv.append(int(ctx[b]) - int(seq_lens[b]) + t + 1)
            rows.append(b)
    want = OK.attention_sinks(q, kc, vc, sinks, bt, torch.tensor(kv, dtype=torch.int32), scale, window, Hq, Hkv, torch.tensor(rows, dtype=torch.int32))
    assert torch.allclose(got.cpu().float(), want.float(), atol=2e-3, rtol=2 ** -6), (got.cpu().float() - want.float()).abs().max()


def _fia_case(total_q, topk1, block_size, Hq, D, dtype, seed):
    """A prefill batch of two requests with scattered pages: every query selects up to topk1 - 1 past blocks (random, -1-padded) and, most of
    the time, its own block somewhere in the list."""
    g = torch.Generator().manual_seed(seed)
    n_req, max_ctx = 2, 6 * block_size
    num_pages = 2 * (max_ctx // block_size) + 3
    perm = torch.randperm(num_pages - 1, generator=g) + 1
    req_to_token = torch.zeros((n_req, max_ctx), dtype=torch.int32)
    for r in range(n_req):
        for b in range(max_ctx // block_size):
            page = int(perm[r * (max_ctx // block_size) + b])
            req_to_token[r, b * block_size:(b + 1) * block_size] = page * block_size + torch.arange(block_size, dtype=torch.int32)
    req = torch.randint(0, n_req, (total_q,), generator=g)
    seq_lens = torch.randint(1, max_ctx + 1, (total_q,), generator=g, dtype=torch.int32)
    seq_lens[0] = 1                                        # the very first position: own block only, one key
    topk = torch.full((total_q, topk1), -1, dtype=torch.int32)
    for t in range(total_q):
        own = (int(seq_lens[t]) - 1) // block_size
        past = torch.randperm(own, generator=g)[:topk1 - 1].tolist() if own > 0 else []
        blocks = past + ([own] if t % 5 != 4 else [])      # every fifth query does not select its own block
        order = torch.randperm(len(blocks), generator=g).tolist()
        for i, j in enumerate(order):
            topk[t, i] = blocks[j]
    k = (torch.randn((num_pages, block_size, 1, D), generator=g)).to(dtype)
    v = (torch.randn((num_pages, block_size, 1, D), generator=g)).to(dtype)
    q = torch.randn((total_q, Hq, D), generator=g).to(dtype)
    return q, k, v, topk, seq_lens, req, req_to_token, num_pages


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("total_q,topk1,block_size,Hq,D", [(37, 4, 16, 16, 128), (64, 16, 64, 16, 128), (9, 3, 128, 8, 64)])
def test_fia_blockq_sparse_prefill(total_q, topk1, block_size, Hq, D, dtype):
    """attention/fia_blockq_attention.py: the per-query block tables and lengths bit for bit against the restated prep kernel (own block last,
    logical -> physical pages, pads 0), the attention against an fp32 softmax over exactly those keys; caller-lent table buffers are used."""
    from sgl_kernel_npu.attention.fia_blockq_attention import flash_prefill_bnsd_blockq_sparse_fia
    q, k, v, topk, seq_lens, req, rtt, num_pages = _fia_case(total_q, topk1, block_size, Hq, D, dtype, total_q * 7 + topk1)
    bt_want, kvl_want = OK.fia_prep(topk, seq_lens, req, rtt, block_size)
    want = OK.fia_blockq_sparse(q, k, v, topk, seq_lens, req, rtt, block_size, D ** -0.5)
    bt = torch.full((total_q, topk1), -7, dtype=torch.int32, device="cuda")
    kvl = torch.full((total_q,), -7, dtype=torch.int32, device="cuda")
    for req_dtype in (torch.int32, torch.int64):
        got = flash_prefill_bnsd_blockq_sparse_fia(q.cuda(), k.cuda(), v.cuda(), topk[None].cuda(), seq_lens.cuda(), req.to(req_dtype).cuda(),
                                                   rtt.cuda(), block_size, None, num_pages, topk1, block_table_out=bt, actual_kvlen_out=kvl)
        assert torch.equal(bt.cpu(), bt_want) and torch.equal(kvl.cpu(), kvl_want)
        tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
        err = (got.cpu().float() - want.float()).abs()
        assert bool((err <= tol * want.float().abs().amax(dim=-1, keepdim=True) + 1e-5).all()), float(err.max())


@pytest.mark.parametrize("B,Hq,Hkv,D,S,page,kind", [(64, 64, 8, 128, 4096, 64, "ragged"), (16, 64, 8, 128, 6000, 16, "ragged"), (8, 256, 1, 128, 3000, 64, "ragged"),
                                                   (24, 32, 4, 64, 9000, 128, "one_long"), (4, 16, 2, 256, 2000, 64, "uniform")])
def test_planned_work_list_and_outputs_match_one_piece_per_sequence(B, Hq, Hkv, D, S, page, kind):
    """decode_gqa with the device-built work list (pieces per sequence by length, longest first) against the same kernel with ONE piece per
    sequence: equal within the fp32 summation-order tolerance of a flash-decoding merge; zero-length sequences included."""
    g = torch.Generator(device="cuda").manual_seed(B + S)
    maxp = (S + page - 1) // page
    nb = B * maxp
    dt = torch.bfloat16
    q = torch.randn((B, Hq, D), generator=g, device="cuda").to(dt)
    k = (torch.randn((nb, page, Hkv, D), generator=g, device="cuda") * 0.5).to(dt)
    v = (torch.randn((nb, page, Hkv, D), generator=g, device="cuda") * 0.5).to(dt)
    bt = torch.randperm(nb, generator=g, device="cuda").to(torch.int32).reshape(B, maxp)
    if kind == "uniform":
        lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    elif kind == "ragged":
        lens = torch.randint(0, S + 1, (B,), generator=g, device="cuda").to(torch.int32)
    else:
        lens = torch.randint(1, 200, (B,), generator=g, device="cuda").to(torch.int32)
        lens[1] = S
    got = run_gqa(q, k, v, lens, bt, D ** -0.5, PLANNED)
    one = run_gqa(q, k, v, lens, bt, D ** -0.5, 1)
    nz = lens.cpu() > 0
    assert not torch.isnan(got.float()[nz]).any()
    assert torch.allclose(got.float()[nz], one.float()[nz], rtol=2 ** -7, atol=2e-3), (got.float()[nz] - one.float()[nz]).abs().max()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("splits", [1, 2, PLANNED])
@pytest.mark.parametrize("view", [False, True], ids=["own_v", "v_view"])
def test_wide_kernel_moves_its_softmax_reference(dtype, splits, view):
    """gqa_decode_wide.hip keeps a LAZY softmax reference per head (it moves only when a tile's maximum exceeds it by > 8 in the log2 domain, the
    head's owner then publishes the accumulator rescale for the dimension owners).  Scores that keep growing along the sequence -- and heads
    that grow at different rates, so that in most tiles only SOME waves move -- must still give the exact softmax."""
    torch.manual_seed(4)
    B, Hq, Lk, Lv, S, page = 2, 128, 288, 256, 640, 64
    maxp = S // page
    q = torch.randn((B, Hq, Lk)).to(dtype)
    k = torch.randn((B * maxp, page, 1, Lk)).to(dtype)
    v = torch.randn((B * maxp, page, 1, Lv)).to(dtype)
    bt = torch.arange(B * maxp, dtype=torch.int32).reshape(B, maxp)
    lens = torch.tensor([S, S - 75], dtype=torch.int32)
    # key n of a sequence gets a component along a fixed direction that grows with n; head h looks along it with weight ~ h
    u = torch.randn(Lk)
    u /= u.norm()
    ramp = torch.linspace(0, 60, S)
    kk = k.float().reshape(B, S, Lk) + ramp[None, :, None] * u[None, None, :]
    k = kk.reshape(B * maxp, page, 1, Lk).to(dtype)
    q = (q.float() + (torch.arange(Hq).float()[None, :, None] / Hq * 6.0) * u[None, None, :]).to(dtype)
    sm = 1.0 / Lk ** 0.5
    if view:                                          # (the 64-key-tile instance: V = the first Lv columns of K)
        v = k[..., :Lv]
    want = OK.decode_gqa(q, k, v, lens, bt, sm)
    kc = k.cuda()
    got = run_gqa(q.cuda(), kc, kc[..., :Lv] if view else v.cuda(), lens.cuda(), bt.cuda(), sm, splits).cpu()
    assert not torch.isnan(got.float()).any()
    check(got, want, exact_fp64(q, k, v, lens, bt, sm), dtype)


@pytest.mark.parametrize("view", [False, True], ids=["own_v", "v_view"])
def test_wide_kernel_full_size_vs_fp32(view):
    """The reference test's own shape at serving size (test_decode_attention.py:242: 128 q heads on 1 kv head, 288 / 256): batch 128, 4096 keys,
    ragged, random page table; sampled rows against an fp32 evaluation on the GPU.  v_view: V = k[..., :256] as the reference test builds it."""
    B, Hq, D, Dv, S, page = 128, 128, 288, 256, 4096, 64
    maxp = S // page
    nb = B * maxp
    g = torch.Generator(device="cuda").manual_seed(6)
    q = torch.randn((B, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn((nb, page, 1, D), generator=g, device="cuda").to(torch.bfloat16)
    v = k[..., :Dv] if view else torch.randn((nb, page, 1, Dv), generator=g, device="cuda").to(torch.bfloat16)
    bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(B, maxp)
    lens = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32)
    lens[0] = S
    for splits in (0, PLANNED):
        got = run_gqa(q, k, v, lens, bt, D ** -0.5, splits)
        for b in (0, 57, 127):
            L = int(lens[b])
            idx = bt[b, :(L + page - 1) // page].long()
            K = k[idx, :, 0].reshape(-1, D)[:L].float()
            V = v[idx, :, 0].reshape(-1, Dv)[:L].float()
            ref = torch.softmax((q[b].float() @ K.T) * D ** -0.5, -1) @ V
            assert torch.allclose(got[b].float(), ref, atol=1e-3, rtol=2 ** -7), (splits, b, (got[b].float() - ref).abs().max())


def _wide_pair_inputs(B, Hq, Lk, Lv, S, page, dtype, ragged, view, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    maxp = (S + page - 1) // page
    nb = B * maxp
    q = torch.randn((B, Hq, Lk), generator=g, device="cuda").to(dtype)
    k = (torch.randn((nb, page, 1, Lk), generator=g, device="cuda") * 0.7).to(dtype)
    v = k[..., :Lv] if view else (torch.randn((nb, page, 1, Lv), generator=g, device="cuda") * 0.7).to(dtype)
    bt = torch.randperm(nb, generator=g, device="cuda").to(torch.int32).reshape(B, maxp)
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    if ragged:
        lens = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32)
    return q, k, v, lens, bt


def _run_gqa_ws(q, k, v, lens, bt, sm, splits, max_len, ws=None):
    B, Hq, Lk = q.shape
    Hkv, Lv = k.shape[2], v.shape[3]
    L = lib()
    if ws is None:
        ws = torch.empty(max(L.mi_gqa_decode_workspace(B, Hq, Lv, splits), 16), dtype=torch.uint8, device=q.device)
    out = torch.full((B, Hq, Lv), float("nan"), dtype=q.dtype, device=q.device)
    rc = L.mi_gqa_decode(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lens), ptr(bt), B, Hq, Hkv, Lk, Lv, k.shape[1], bt.stride(0), max_len,
                         q.stride(0), q.stride(1), k.stride(0), k.stride(1), k.stride(2), v.stride(0), v.stride(1), v.stride(2),
                         out.stride(0), out.stride(1), sm, 0 if q.dtype == torch.bfloat16 else 1, splits, ptr(ws), ws.numel(), stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out, ws


# B, Hq, Lk, Lv, S, page, dtype, ragged, view, num_splits -- batches the uniform form (num_splits = 2: what the library picks for the reference
# shape at batch 128) or the work list cuts into two pieces per sequence: the reference test's shape at a quarter of its length, a ragged
# copy (lists with sequences of one, two and three pieces side by side), a partly filled second head half in fp16 with V a cache of its own and
# 32-key pages, narrower head dims, and more workgroups than CUs (a piece may meet a partner that is not resident yet: the bounded wait, then the
# merge kernel -- or simply a late partner)
WIDE_PAIR_CASES = [(128, 128, 288, 256, 1024, 64, torch.bfloat16, False, True, 2), (128, 128, 288, 256, 1500, 64, torch.bfloat16, True, True, 2),
                   (128, 128, 288, 256, 1500, 64, torch.bfloat16, True, True, PLANNED), (100, 96, 288, 256, 700, 32, torch.float16, True, False, PLANNED),
                   (7, 72, 256, 192, 900, 128, torch.float16, True, False, 2), (200, 128, 288, 256, 700, 64, torch.bfloat16, True, True, 2),
                   (200, 128, 288, 256, 1100, 64, torch.float16, True, True, PLANNED)]


@pytest.mark.parametrize("B,Hq,Lk,Lv,S,page,dtype,ragged,view,splits", WIDE_PAIR_CASES)
def test_wide_kernel_two_piece_sequences_finish_between_their_workgroups(B, Hq, Lk, Lv, S, page, dtype, ragged, view, splits):
    """Sequences in two pieces finish inside gqa_decode_wide.hip (each workgroup: 64 of the group's heads) with the merge kernel's sums in the
    merge kernel's order: the outputs are THE BITS of the run with the pair finish off -- also when the second piece withholds its word and the
    first runs into its bounded wait (mode 2: the merge kernel does the work), and over repeated calls on one workspace (the meeting words are
    re-armed by the merge kernel)."""
    L = lib()
    L.mi_gqa_decode_set_pair.argtypes = [c_int]
    q, k, v, lens, bt = _wide_pair_inputs(B, Hq, Lk, Lv, S, page, dtype, ragged, view, 5)
    sm = Lk ** -0.5
    try:
        assert L.mi_gqa_decode_set_pair(0) == 0
        want, _ = _run_gqa_ws(q, k, v, lens, bt, sm, splits, S)
        assert not torch.isnan(want.float()).any()
        assert L.mi_gqa_decode_set_pair(1) == 0
        got, ws = _run_gqa_ws(q, k, v, lens, bt, sm, splits, S)
        assert torch.equal(got, want), (got.float() - want.float()).abs().max()
        for rep in range(3):                                   # the same workspace again and again, with other queries in between
            q2 = (q.float() * (1.0 + 0.25 * rep)).to(dtype)
            L.mi_gqa_decode_set_pair(0)
            want2, _ = _run_gqa_ws(q2, k, v, lens, bt, sm, splits, S)
            L.mi_gqa_decode_set_pair(1)
            got2, _ = _run_gqa_ws(q2, k, v, lens, bt, sm, splits, S, ws=ws)
            assert torch.equal(got2, want2), rep
        assert L.mi_gqa_decode_set_pair(2) == 0
        got, _ = _run_gqa_ws(q, k, v, lens, bt, sm, splits, S, ws=ws)
        assert torch.equal(got, want), "bounded wait -> merge kernel"
        assert L.mi_gqa_decode_set_pair(1) == 0                # the words a timed-out call left behind do not leak into the next one
        got, _ = _run_gqa_ws(q, k, v, lens, bt, sm, splits, S, ws=ws)
        assert torch.equal(got, want)
    finally:
        L.mi_gqa_decode_set_pair(-1)


def test_wide_kernel_pair_finish_in_a_replayed_graph():
    """The pair finish of gqa_decode_wide.hip inside a captured graph: every replay carries the same tag, the meeting words are re-armed by the
    merge kernel of the replay before -- outputs follow the inputs written between replays, bit for bit the eager result with the finish off."""
    import sgl_kernel_npu.attention.decode_attention  # noqa: F401  (registers torch.ops.npu.decode_gqa)
    L = lib()
    L.mi_gqa_decode_set_pair.argtypes = [c_int]
    B, Hq, D, Dv, S, page = 128, 128, 288, 256, 1024, 64
    q, k, v, lens, bt = _wide_pair_inputs(B, Hq, D, Dv, S, page, torch.bfloat16, False, True, 11)
    out = torch.empty((B, Hq, Dv), dtype=torch.bfloat16, device="cuda")
    try:
        L.mi_gqa_decode_set_pair(1)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            torch.ops.npu.decode_gqa(q, k, v, out, lens, D ** -0.5, page, bt, 0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            torch.ops.npu.decode_gqa(q, k, v, out, lens, D ** -0.5, page, bt, 0)
        for rep in range(4):
            q.copy_((torch.randn(q.shape, device="cuda") * (1 + rep)).to(torch.bfloat16))
            out.fill_(7.0)
            graph.replay()
            torch.cuda.synchronize()
            got = out.clone()
            L.mi_gqa_decode_set_pair(0)
            want = torch.empty_like(out)
            torch.ops.npu.decode_gqa(q, k, v, want, lens, D ** -0.5, page, bt, 0)
            torch.cuda.synchronize()
            L.mi_gqa_decode_set_pair(1)
            assert torch.equal(got, want), rep
    finally:
        L.mi_gqa_decode_set_pair(-1)
