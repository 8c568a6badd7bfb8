"""GPU parity of the fused elementwise primitives (through torch.ops.npu -> C-ABI) against the CPU oracle, with the
tolerances of the reference's own tests."""
import numpy as np
import pytest
import torch

from oracle import kernels as OK

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _load():
    import sgl_kernel_npu  # noqa: F401


@pytest.mark.parametrize("s,h", [(4096, 3072), (300, 4096), (64, 8192), (7, 32)])
@pytest.mark.parametrize("gl_dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("gl_type", [1, 0])
def test_swiglu_quant(s, h, gl_dtype, gl_type):
    from sgl_kernel_npu.activation.swiglu_quant import swiglu_quant
    torch.manual_seed(1)
    x = (torch.randn((s, h)) * 2).to(torch.bfloat16)
    counts = torch.tensor([0, 32, 0, 0, 10, 0, 0, 0, 100, 0, 0, 5, 5, 5, 0, 0], dtype=gl_dtype)
    if int(counts.sum()) > s:
        counts = torch.tensor([1, 0, s - 3, 0], dtype=gl_dtype)
    gl = counts if gl_type == 1 else torch.cumsum(counts, 0).to(gl_dtype)
    want_q, want_s, total = OK.swiglu_quant(x, gl, gl_type)
    q, sc = swiglu_quant(x.cuda(), gl.cuda(), gl_type)
    d = (q[:total].cpu().int() - want_q[:total].int()).abs()
    assert d.max() <= 1 and (d > 0).float().mean() < 2e-2            # test_swiglu_quant.py:45-54
    assert torch.allclose(sc[:total].cpu(), want_s[:total], rtol=5e-3)
    # un-quantised and clamped variants
    o, _ = swiglu_quant(x.cuda(), gl.cuda(), gl_type, need_quant=False)
    wo, _, _ = OK.swiglu_quant(x, gl, gl_type, need_quant=False)
    assert torch.allclose(o[:total].cpu().float(), wo[:total].float(), rtol=2e-2, atol=2e-2)
    q3, s3 = swiglu_quant(x.cuda(), gl.cuda(), gl_type, do_limit=True, limit=1.5)
    w3, ws3, _ = OK.swiglu_quant(x, gl, gl_type, do_limit=True, limit=1.5)
    assert (q3[:total].cpu().int() - w3[:total].int()).abs().max() <= 1
    with pytest.raises(ValueError):
        swiglu_quant(x.cuda(), gl.cuda(), 2)


@pytest.mark.parametrize("B,H", [(3, 6144), (128, 7168), (1, 8192), (5, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_add_rmsnorm_bias(B, H, dtype):
    from sgl_kernel_npu.norm.add_rmsnorm_bias import add_gemma_rms_norm, add_rmsnorm_bias
    torch.manual_seed(0)
    x, r = torch.randn(B, H).to(dtype), torch.randn(B, H).to(dtype)
    w, b = torch.randn(H).to(dtype), torch.randn(H).to(dtype)
    w1, w2 = OK.add_rmsnorm_bias(x, r, w, b, 1e-6)
    o1, o2 = add_rmsnorm_bias(x.cuda(), r.cuda(), w.cuda(), b.cuda(), 1e-6)
    assert torch.equal(o2.cpu(), w2)                                              # the sum is exact in the I/O dtype
    ulp = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    assert torch.allclose(o1.cpu().float(), w1.float(), rtol=ulp, atol=1e-3)      # reference: rtol 5e-3
    qs, qo = torch.randn(H).to(dtype), torch.randn(H).to(dtype)
    wq, _ = OK.add_rmsnorm_bias(x, r, w, b, 1e-6, qs, qo)
    oq, _ = add_rmsnorm_bias(x.cuda(), r.cuda(), w.cuda(), b.cuda(), 1e-6, qs.cuda(), qo.cuda())
    assert oq.dtype == torch.int8
    d = (oq.cpu().int() - wq.int()).abs()
    assert d.max() <= 1 and (d > 0).float().mean() < 1e-2
    gn, ga = OK.add_rmsnorm_bias(x, r, w, None, 1e-6, gemma=True)
    n, a = add_gemma_rms_norm(x.cuda(), w.cuda(), r.cuda(), 1e-6)
    assert torch.equal(a.cpu(), ga) and torch.allclose(n.cpu().float(), gn.float(), rtol=ulp, atol=1e-3)
    from sgl_kernel_npu.norm.rmsnorm_bias import rmsnorm_bias
    p1 = rmsnorm_bias(x.cuda(), w.cuda(), b.cuda(), 1e-6)                         # no residual (rmsnorm_bias.py:78-120)
    wp1, _ = OK.add_rmsnorm_bias(x, None, w, b, 1e-6)
    assert torch.allclose(p1.cpu().float(), wp1.float(), rtol=ulp, atol=1e-3)
    n2, a2 = add_gemma_rms_norm(x.cuda(), w.cuda(), None, 1e-6)                  # no residual
    gn2, _ = OK.add_rmsnorm_bias(x, None, w, None, 1e-6, gemma=True)
    assert torch.allclose(n2.cpu().float(), gn2.float(), rtol=ulp, atol=1e-3)


@pytest.mark.parametrize("rope_dim,neox,norm,bias", [(128, True, True, True), (64, True, True, True), (128, False, True, False),
                                                     (64, False, False, False), (32, True, True, False)])
@pytest.mark.parametrize("hd,qh,kvh,B", [(128, 6144, 1024, 12), (64, 512, 128, 3), (256, 1024, 256, 5)])
def test_split_qkv_rmsnorm_rope(rope_dim, neox, norm, bias, hd, qh, kvh, B):
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkv_rmsnorm_rope
    if rope_dim > hd:
        pytest.skip("rope_dim > head_dim")
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    qkv = torch.randn(B, qh + 2 * kvh).to(torch.bfloat16)
    qw, kw, qb, kb = [torch.randn(hd).to(torch.bfloat16) for _ in range(4)]
    sin = torch.from_numpy(rng.uniform(0, 1, [B, 1, 1, rope_dim])).to(torch.bfloat16)
    cos = torch.from_numpy(rng.uniform(0, 1, [B, 1, 1, rope_dim])).to(torch.bfloat16)
    kwargs = dict(eps=1e-6 if norm else None, q_weight=qw if norm else None, k_weight=kw if norm else None,
                  q_bias=qb if (norm and bias) else None, k_bias=kb if (norm and bias) else None, is_neox_style=neox)
    wq, wk, wv = OK.split_qkv_rmsnorm_rope(qkv, sin, cos, qh, kvh, hd, **kwargs)
    dev = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
    q, k, v = split_qkv_rmsnorm_rope(qkv.cuda(), sin.cuda(), cos.cuda(), qh, kvh, hd, **dev)
    assert torch.equal(v.cpu(), wv)
    # fp32 math on both sides: at most one bf16 ulp apart (reference tolerance: atol 5e-2)
    assert torch.allclose(q.cpu().float(), wq.float(), rtol=2 ** -7, atol=1e-3)
    assert torch.allclose(k.cpu().float(), wk.float(), rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("hd,qh,kvh,B,rope_dim", [(128, 2048, 512, 7, 64), (128, 4096, 1024, 33, 128), (64, 256, 64, 3, 64), (256, 1024, 256, 4, 32),
                                                  (8, 16, 8, 2, 4), (1024, 2048, 1024, 2, 512)])
def test_split_qkvgate_gemma_rmsnorm_rope(hd, qh, kvh, B, rope_dim, dt):
    """split_qkv_rmsnorm_rope.py:441-745 through torch.ops.npu: gate and V bit for bit, q and k within one ulp of the fp32 restatement."""
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkvgate_gemma_rmsnorm_rope
    torch.manual_seed(hd * 3 + B)
    x = torch.randn(B, 2 * qh + 2 * kvh).to(dt)
    qw, kw = torch.randn(hd).to(dt), torch.randn(hd).to(dt)
    sin, cos = torch.rand(B, rope_dim).to(dt), torch.rand(B, rope_dim).to(dt)
    wq, wk, wv, wg = OK.split_qkvgate_gemma_rmsnorm_rope(x, sin, cos, qh, kvh, hd, rope_dim, 1e-6, qw, kw)
    q, k, v, g = split_qkvgate_gemma_rmsnorm_rope(x.cuda(), sin.cuda(), cos.cuda(), qh, kvh, hd, rope_dim, 1e-6, qw.cuda(), kw.cuda())
    assert q.shape == (B, qh) and k.shape == (B, kvh) and v.shape == (B, kvh) and g.shape == (B, qh)
    assert torch.equal(v.cpu(), wv) and torch.equal(g.cpu(), wg)
    ulp = 2 ** -7 if dt == torch.bfloat16 else 2 ** -10
    assert torch.allclose(q.cpu().float(), wq.float(), rtol=ulp, atol=1e-3)
    assert torch.allclose(k.cpu().float(), wk.float(), rtol=ulp, atol=1e-3)
    assert (q.cpu().view(torch.int16) != wq.view(torch.int16)).float().mean() < 0.02      # fp32 on both sides: rounding-boundary cases only


def test_split_qkvgate_gemma_rmsnorm_rope_rejects_bad_shapes():
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkvgate_gemma_rmsnorm_rope
    x = torch.randn(2, 2 * 256 + 2 * 128, device="cuda").bfloat16()
    w = torch.randn(128, device="cuda").bfloat16()
    sc = torch.rand(2, 64, device="cuda").bfloat16()
    with pytest.raises(RuntimeError):
        split_qkvgate_gemma_rmsnorm_rope(x[:, :-8].contiguous(), sc, sc, 256, 128, 128, 64, 1e-6, w, w)       # input width
    with pytest.raises(AssertionError):
        split_qkvgate_gemma_rmsnorm_rope(x, sc, sc, 256, 96, 96, 64, 1e-6, w, w)                                # head_dim not a power of two


_ROW_DT = [torch.float32, torch.bfloat16, torch.float16]
_ROW_TOL = {torch.float32: 2e-6, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10}


@pytest.mark.parametrize("dt", _ROW_DT)
@pytest.mark.parametrize("B,C", [(2048, 8), (5, 32), (33, 40), (7, 4096), (3, 20000), (1, 1)])
def test_l1_norm(B, C, dt):
    """norm/l1_norm.py against the reference test's golden (test_l1_norm.py: 2048 x 8 bf16, rtol 5e-3): fp32 out."""
    from sgl_kernel_npu.norm.l1_norm import l1_norm
    torch.manual_seed(B + C)
    x = (torch.rand(B, C) + 0.1).to(dt)                  # positive rows: the sum is well away from zero
    got = l1_norm(x.cuda())
    want = OK.l1_norm(x)
    assert got.dtype == torch.float32 and got.shape == (B, C)
    assert torch.allclose(got.cpu(), want, rtol=2e-5, atol=0)       # fp32 on both sides: summation order only
    assert torch.allclose(got.cpu().sum(-1), torch.ones(B), atol=1e-4)


@pytest.mark.parametrize("dt", _ROW_DT)
@pytest.mark.parametrize("B,L,C", [(1, 130, 2048), (2, 5, 24), (1, 3, 6000), (1, 9, 20008), (4, 1, 4)])
def test_rmsnorm_without_weight(B, L, C, dt):
    """norm/rmsnorm_without_weight.py against F.rms_norm (the reference test's golden: 1 x 130 x 2048 fp32, rtol 1e-3)."""
    from sgl_kernel_npu.norm.rmsnorm_without_weight import fused_rmsnorm_without_weight
    torch.manual_seed(L + C)
    x = torch.randn(B, L, C).to(dt)
    got = fused_rmsnorm_without_weight(x.cuda(), 1e-6)
    want = OK.rmsnorm_without_weight(x, 1e-6)
    assert got.dtype == dt and got.shape == x.shape
    assert torch.allclose(got.cpu().float(), want.float(), rtol=_ROW_TOL[dt], atol=1e-6)
    if dt == torch.float32:
        assert torch.allclose(got.cpu(), torch.nn.functional.rms_norm(x, (C,), eps=1e-6), rtol=1e-3)      # the reference's own assertion


@pytest.mark.parametrize("dt", _ROW_DT)
@pytest.mark.parametrize("B,L,C", [(1, 1024, 512), (1, 8190, 2560), (2, 7, 36), (1, 5, 20008)])
def test_rmsnorm_split(B, L, C, dt):
    """norm/rmsnorm_split.py: fused_variance and fused_rsqrt_mul against the reference test's goldens (fp32, rtol 1e-3)."""
    from sgl_kernel_npu.norm.rmsnorm_split import fused_rsqrt_mul, fused_variance
    torch.manual_seed(L)
    x = torch.randn(B, L, C).to(dt)
    w = torch.randn(C).to(dt)
    var = (torch.randn(1, B * L, 1).abs() + 0.1).to(dt)
    gv = fused_variance(x.cuda())
    assert gv.shape == (B, L, 1) and gv.dtype == dt
    assert torch.allclose(gv.cpu().float(), OK.fused_variance(x).float(), rtol=max(_ROW_TOL[dt], 1e-5), atol=1e-7)
    go = fused_rsqrt_mul(x.cuda(), var.view(-1).cuda(), w.cuda(), 1e-6)
    want = OK.fused_rsqrt_mul(x, var.view(-1), w, 1e-6)
    assert go.shape == x.shape and go.dtype == dt
    assert torch.allclose(go.cpu().float(), want.float(), rtol=_ROW_TOL[dt] * 2, atol=1e-6)
    if dt == torch.float32:
        assert torch.allclose(go.cpu(), x * torch.rsqrt(var.reshape(B, L, 1) + 1e-6) * w, rtol=1e-3)     # the reference's own assertion


@pytest.mark.parametrize("dt,ss", [(torch.float32, torch.float32), (torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32),
                                   (torch.float16, torch.float16)])
@pytest.mark.parametrize("B,L,C", [(3, 130, 5120), (2, 7, 36), (1, 5, 33), (1, 1, 64)])
@pytest.mark.parametrize("form", ["per_column", "scalars", "per_element", "per_element_constant"])
def test_fused_scale_shift(B, L, C, form, dt, ss):
    """norm/scale_shift.py against the reference test's golden x * (1 + scale) + shift, its three scale / shift shapes (test_scale_shift.py:
    19-32; 3 x 37440 x 5120 there) and the scale_constant of the per-element form."""
    from sgl_kernel_npu.norm.scale_shift import fused_scale_shift
    torch.manual_seed(C)
    x = torch.randn(B, L, C).to(dt)
    c = 1.0
    if form == "per_column":
        scale, shift = torch.randn(1, 1, C), torch.randn(1, 1, C)
    elif form == "scalars":
        scale, shift = torch.randn(1), torch.randn(1)
    else:
        scale, shift = torch.randn(1, 1, C), torch.randn(B, L, C)
        c = 0.5 if form == "per_element_constant" else 1.0
    scale, shift = scale.to(ss), shift.to(ss)
    got = fused_scale_shift(x.cuda(), scale.cuda(), shift.cuda(), c)
    want = OK.fused_scale_shift(x, scale, shift, c)
    assert got.dtype == dt and got.shape == x.shape
    tol = {torch.float32: 1e-6, torch.bfloat16: 2 ** -8, torch.float16: 2 ** -11}[dt]
    assert torch.allclose(got.cpu().float(), want.float(), rtol=tol, atol=tol)
    if dt == torch.float32 and c == 1.0:
        assert torch.allclose(got.cpu(), x * (1 + scale) + shift, rtol=1e-3, atol=1e-6)      # the reference's own assertion


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("qh,kvh,hd,B,rot", [(6144, 1024, 128, 12, 128), (512, 128, 64, 3, 32), (256, 256, 256, 5, 64), (1024, 512, 128, 1, 128)])
def test_split_qkv_tp_rmsnorm_rope(qh, kvh, hd, B, rot, dt):
    """norm/split_qkv_tp_rmsnorm_rope.py at tp_world = 1 against the reference test's golden (6144 / 1024 / 128, 12 rows, bf16, atol 5e-2 there):
    V bit for bit; q, k within one ulp of the golden's two roundings (the sum of squares is added up in another order)."""
    from sgl_kernel_npu.norm.split_qkv_tp_rmsnorm_rope import split_qkv_tp_rmsnorm_rope
    torch.manual_seed(qh + B)
    qkv = torch.randn(B, qh + 2 * kvh).to(dt)
    qw, kw = torch.randn(qh).to(dt), torch.randn(kvh).to(dt)
    sin, cos = torch.rand(B, rot).to(dt), torch.rand(B, rot).to(dt)
    q, k, v = split_qkv_tp_rmsnorm_rope(qkv.cuda(), cos.cuda(), sin.cuda(), qh, kvh, hd, 1e-6, qw.cuda(), kw.cuda(), rot, 1, None)
    wq, wk, wv = OK.split_qkv_tp_rmsnorm_rope(qkv, cos, sin, qh, kvh, hd, 1e-6, qw, kw, rot)
    assert torch.equal(v.cpu(), wv)
    ulp = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10, torch.float32: 1e-5}[dt]
    for got, want in ((q, wq), (k, wk)):
        assert got.dtype == dt
        assert torch.allclose(got.cpu().float(), want.float(), rtol=2 * ulp, atol=2 * ulp)       # two roundings: norm, then rotation
        assert torch.allclose(got.cpu().float(), want.float(), atol=5e-2)                        # the reference's own bar


def test_split_qkv_tp_rmsnorm_rope_two_launch_form_with_a_foreign_variance():
    """tp_world = 2 without a second rank: the second launch is fed local + foreign variance, as the all-reduce would."""
    torch.manual_seed(5)
    B, qh, kvh, hd = 6, 512, 128, 64
    dt = torch.bfloat16
    qkv = torch.randn(B, qh + 2 * kvh).to(dt)
    qw, kw = torch.randn(qh).to(dt), torch.randn(kvh).to(dt)
    sin, cos = torch.rand(B, hd).to(dt), torch.rand(B, hd).to(dt)
    other = torch.rand(B, 2) + 0.5
    v, var = torch.ops.npu.split_qkv_tp_local_var(qkv.cuda(), qh, kvh)
    want_var = torch.stack([qkv[:, :qh].float().pow(2).mean(-1), qkv[:, qh:qh + kvh].float().pow(2).mean(-1)], dim=-1)
    assert torch.allclose(var.cpu(), want_var, rtol=1e-5)
    q, k = torch.ops.npu.split_qkv_tp_norm_rope(qkv.cuda(), cos.cuda(), sin.cuda(), var + other.cuda(), qh, kvh, hd, 1e-6, qw.cuda(), kw.cuda(), hd, 0.5)
    wq, wk, wv = OK.split_qkv_tp_rmsnorm_rope(qkv, cos, sin, qh, kvh, hd, 1e-6, qw, kw, hd, tp_world=2, other_var=other)
    assert torch.equal(v.cpu(), wv)
    assert torch.allclose(q.cpu().float(), wq.float(), rtol=2 ** -6, atol=2 ** -6) and torch.allclose(k.cpu().float(), wk.float(), rtol=2 ** -6, atol=2 ** -6)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,nq,nkv,hd,rope_dim,sections,inter,gate,bias", [
    (7, 2, 1, 256, 256, [32, 48, 48], True, True, False),         # the reference test's two cases (test_split_qkv_rmsnorm_mrope.py:113-122)
    (65, 2, 1, 256, 128, [48, 40, 40], False, False, True),
    (33, 16, 2, 128, 128, [24, 20, 20], True, False, False),      # Qwen2.5-VL / Qwen3-VL head shapes
    (33, 16, 2, 128, 128, [16, 24, 24], False, True, True),
    (33, 16, 2, 128, 128, [24, 20, 20], False, False, True),      # the branch-free instance (heads of 128 rotated whole, not gated) with biases
    (5, 4, 4, 64, 32, [8, 4, 4], False, False, False),
    (9, 4, 2, 128, 128, [16, 16, 16], False, False, False),       # sections end before rope_dim / 2: offsets 48..63 take cos = sin = 0 (kernel :157-163)
])
def test_split_qkv_rmsnorm_mrope(T, nq, nkv, hd, rope_dim, sections, inter, gate, bias, dt):
    """norm/split_qkv_rmsnorm_mrope.py against the reference test's golden (its two cases and more): V and gate bit for bit, q and k within one
    output ulp of the golden's fp32 evaluation (the reference's own bar: atol 5e-2, rtol 5e-3)."""
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_mrope import triton_split_qkv_rmsnorm_mrope, triton_split_qkv_rmsnorm_mrope_fake
    torch.manual_seed(T + hd)
    qs, kvs = nq * hd, nkv * hd
    qkv = torch.randn(T, qs + (qs if gate else 0) + 2 * kvs).to(dt)
    qw, kw = torch.randn(hd).to(dt), torch.randn(hd).to(dt)
    qb, kb = (torch.randn(hd).to(dt), torch.randn(hd).to(dt)) if bias else (None, None)
    cos_sin = torch.randn(3, T, rope_dim).to(dt)
    want = OK.split_qkv_rmsnorm_mrope(qkv, qw, kw, cos_sin, nq, nkv, hd, 1e-6, sections, inter, rope_dim, qb, kb, gate)
    c = lambda t: None if t is None else t.cuda()
    got = triton_split_qkv_rmsnorm_mrope(qkv.cuda(), qw.cuda(), kw.cuda(), cos_sin.cuda(), nq, nkv, hd, 1e-6, sections, inter, rope_dim, c(qb), c(kb), gate)
    fake = triton_split_qkv_rmsnorm_mrope_fake(qkv, qw, kw, cos_sin, nq, nkv, hd, 1e-6, sections, inter, rope_dim, qb, kb, gate)
    for g_, w_, f_ in zip(got, want, fake):
        assert g_.shape == w_.shape == f_.shape and g_.dtype == dt
    assert torch.equal(got[2].cpu(), want[2]) and torch.equal(got[3].cpu(), want[3])
    ulp = 2 ** -7 if dt == torch.bfloat16 else 2 ** -10
    for i in (0, 1):
        assert torch.allclose(got[i].cpu().float(), want[i].float(), rtol=ulp, atol=2e-3)
        torch.testing.assert_close(got[i].cpu().float(), want[i].float(), atol=5e-2, rtol=5e-3)      # the reference's own assertion


@pytest.mark.parametrize("cache_dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,qh,kvh,hd,rope_dim,norm,bias,cast,pos_dt", [
    (12, 2048, 512, 128, 128, True, False, True, torch.int64),      # the reference test's shapes (full and partial rotary, with and without qk norm)
    (12, 2048, 512, 128, 64, True, False, True, torch.int64),
    (12, 2048, 512, 128, 128, False, False, True, torch.int64),
    (1500, 1024, 256, 128, 128, True, True, True, torch.int32),     # past the reference's wide-grid threshold; bias
    (7, 512, 512, 256, 32, True, True, False, torch.int32),         # no rounding between norm and rotation
    (3, 256, 64, 64, 64, True, False, True, torch.int64),
])
def test_split_qkv_rmsnorm_rope_pos_cache_half(B, qh, kvh, hd, rope_dim, norm, bias, cast, pos_dt, cache_dt):
    """norm/split_qkv_rmsnorm_rope_pos_cache_half_npu.py: V bit for bit; q, k within one output ulp of the kernel's restatement and within the
    reference test's bar (atol 5e-2, rtol 5e-3); out-of-range positions clamp."""
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope_pos_cache_half_npu import split_qkv_rmsnorm_rope_pos_cache_half_npu
    torch.manual_seed(B + hd)
    dt = torch.bfloat16
    max_pos = 2048
    qkv = torch.randn(B, qh + 2 * kvh).to(dt)
    pos = torch.randint(0, max_pos, (B,), dtype=pos_dt)
    pos[0] = -3
    if B > 1:
        pos[1] = max_pos + 7
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, rope_dim, 2, dtype=torch.float32) / rope_dim))
    freqs = torch.einsum("i,j -> ij", torch.arange(max_pos, dtype=torch.float32), inv_freq)
    cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(cache_dt)
    qw, kw = torch.randn(hd + 3).to(dt), torch.randn(hd).to(dt)          # "at least head_dim elements"
    qb, kb = (torch.randn(hd).to(dt), torch.randn(hd).to(dt)) if bias else (None, None)
    c = lambda t: None if t is None else t.cuda()
    kwargs = dict(eps=1e-6 if norm else None, q_weight=c(qw) if norm else None, k_weight=c(kw) if norm else None, q_bias=c(qb), k_bias=c(kb),
                  rope_dim=rope_dim, cast_norm_to_bf16=cast)
    q, k, v = split_qkv_rmsnorm_rope_pos_cache_half_npu(qkv.cuda(), pos.cuda(), cache.cuda(), qh, kvh, hd, **kwargs)
    wq, wk, wv = OK.split_qkv_rmsnorm_rope_pos_cache_half(qkv, pos, cache, qh, kvh, hd, 1e-6 if norm else None, qw if norm else None,
                                                          kw if norm else None, qb, kb, rope_dim, cast)
    assert torch.equal(v.cpu(), wv)
    for got, want in ((q, wq), (k, wk)):
        assert torch.allclose(got.cpu().float(), want.float(), rtol=2 ** -7, atol=2e-3)
        torch.testing.assert_close(got.cpu().float(), want.float(), atol=5e-2, rtol=5e-3)
        assert (got.cpu().view(torch.int16) != want.view(torch.int16)).float().mean() < 0.02


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,ql,kl,rd,qbias,kbias", [(128, 1536, 512, 64, False, False), (5, 1536, 512, 64, True, True), (3, 100, 36, 10, True, False)])
def test_fused_split_qk_norm(B, ql, kl, rd, qbias, kbias, dt):
    """norm/fused_split_qk_norm.py (the MLA down-projection split, DeepSeek shapes 1536 / 512 / 64): rope part bit for bit, the two normed parts
    within one output ulp of the fp32 restatement; layer-norm modules with and without a bias."""
    from sgl_kernel_npu.norm.fused_split_qk_norm import fused_split_qk_norm
    torch.manual_seed(B)
    x = torch.randn(B, ql + kl + rd).to(dt)

    class LN:                                       # stands in for the model's RMSNorm modules: .weight, optionally .bias
        pass
    qln, kln = LN(), LN()
    qln.weight, kln.weight = torch.randn(ql).to(dt).cuda(), torch.randn(kl).to(dt).cuda()
    if qbias:
        qln.bias = torch.randn(ql).to(dt).cuda()
    if kbias:
        kln.bias = torch.randn(kl).to(dt).cuda()
    q, kn, kp = fused_split_qk_norm(x.cuda(), qln, kln, ql, kl, rd, 1e-6)
    cpu = lambda t: None if t is None else t.cpu()
    wq, wkn, wkp = OK.fused_split_qk_norm(x, cpu(qln.weight), cpu(getattr(qln, "bias", None)), cpu(kln.weight), cpu(getattr(kln, "bias", None)), ql, kl, rd, 1e-6)
    assert q.shape == wq.shape and kn.shape == wkn.shape == (B, 1, kl) and kp.shape == wkp.shape == (B, 1, rd)
    assert torch.equal(kp.cpu(), wkp)
    ulp = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10, torch.float32: 2e-6}[dt]
    assert torch.allclose(q.cpu().float(), wq.float(), rtol=ulp, atol=1e-5) and torch.allclose(kn.cpu().float(), wkn.float(), rtol=ulp, atol=1e-5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,dim", [(1024, 5760), (7, 64), (1, 16), (300, 2880 * 2)])
def test_swiglu_oai(rows, dim, dt):
    """activation/swiglu_oai.py (GPT-OSS: 2880 intermediate columns) through the module-style entry point the reference exposes."""
    from sgl_kernel_npu.activation.swiglu_oai import swiglu_oai, swiglu_oai_native
    torch.manual_seed(rows)
    x = (torch.randn(rows, dim) * 4).to(dt)

    class Cfg:
        gemm1_alpha, gemm1_clamp_limit = 1.702, 7.0

    class Layer:
        w13_weight = torch.empty(2, 1, dim)
        moe_runner_config = Cfg()
    got = swiglu_oai(Layer(), x.cuda())
    want = OK.swiglu_oai(x, dim, 1.702, 7.0)
    assert got.shape == (rows, dim // 2) and got.dtype == dt
    ulp = {torch.bfloat16: 2 ** -7, torch.float16: 2 ** -10, torch.float32: 1e-5}[dt]
    assert torch.allclose(got.cpu().float(), want.float(), rtol=2 * ulp, atol=1e-5)
    native = swiglu_oai_native(Layer(), x)                       # dtype arithmetic: several roundings
    assert torch.allclose(got.cpu().float(), native.float(), rtol=6 * ulp + 1e-5, atol=6 * ulp)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,cols,mode", [(300, 5760, "dense"), (64, 128, "dense"), (257, 2880 * 2, "counts"), (100, 512, "cumsum"), (9, 6, "noquant"), (9, 6, "dense"), (12, 16384, "dense"),
                                            (12, 16384, "noquant")])
def test_swiglu_oai_quant(rows, cols, mode, dt):
    """activation/swiglu_oai_quant.py: dense and grouped forms.  fp32 on both sides but exp / reciprocal differ in the last bit, and the value is
    rounded to the I/O dtype before the truncating cast, so a step may move at a rounding boundary: |dq| <= 1 on < 2 % of the elements (the bar
    the reference sets for its sibling swiglu_quant, test_swiglu_quant.py:45-54), scales to 1e-5, and the dequantised row within one step."""
    from sgl_kernel_npu.activation.swiglu_oai_quant import swiglu_oai_quant
    torch.manual_seed(rows + cols)
    x = (torch.randn(rows, cols) * 3).to(dt)
    alpha, limit = 1.702, 7.0
    gl, glt, total = None, None, rows
    if mode == "counts":
        gl, glt = torch.tensor([100, 0, 57, 60], dtype=torch.int64), 1
        total = 217
    elif mode == "cumsum":
        gl, glt = torch.tensor([10, 10, 64, 90], dtype=torch.int32), 0
        total = 90
    need_quant = mode != "noquant"
    got, gs = swiglu_oai_quant(x.cuda(), alpha, limit, need_quant, None if gl is None else gl.cuda(), glt)
    want, ws = OK.swiglu_oai_quant(x, alpha, limit, need_quant, total)
    assert got.shape == (rows, cols // 2) and gs.shape == (rows,)
    if not need_quant:
        assert got.dtype == dt
        assert torch.allclose(got.cpu().float(), want.float(), rtol=2 ** -7 if dt == torch.bfloat16 else 2 ** -10, atol=1e-5)
        return
    assert got.dtype == torch.int8
    assert torch.allclose(gs.cpu()[:total], ws[:total], rtol=1e-5)
    d = (got.cpu()[:total].int() - want[:total].int()).abs()
    assert d.max() <= 1 and (d != 0).float().mean() < 2e-2
    with pytest.raises(ValueError):
        swiglu_oai_quant(x.cuda(), alpha, limit, True, torch.tensor([1, 2]).cuda(), 3)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,cols,mode,linear_beta", [(300, 6144, "dense", 25.0), (64, 128, "dense", None), (257, 12288, "counts", 25.0),
                                                        (100, 512, "cumsum", 3.0), (9, 6, "dense", 25.0), (40, 2 * 8448, "noquant", 25.0)])
def test_situ(rows, cols, mode, linear_beta, dt):
    """activation/situ.py: situ_and_mul, situ_and_mul_quant and situ against the fp32 restatement.  tanh / exp differ from torch's in the last
    bits, so a quantised value may move by one step at a rounding boundary: |dq| <= 1 on < 2 % of the elements, scales to 1e-5; the
    unquantised output to one rounding of the I/O dtype."""
    from sgl_kernel_npu.activation.situ import situ, situ_and_mul, situ_and_mul_quant
    torch.manual_seed(rows + cols)
    x = (torch.randn(rows, cols) * 3).to(dt)
    beta = 4.0
    gl, glt, total = None, None, rows
    if mode == "counts":
        gl, glt, total = torch.tensor([100, 0, 57, 60], dtype=torch.int64), 1, 217
    elif mode == "cumsum":
        gl, glt, total = torch.tensor([10, 10, 64, 90], dtype=torch.int32), 0, 90
    glc = None if gl is None else gl.cuda()
    tol = dict(rtol=2 ** -7 if dt == torch.bfloat16 else 2 ** -10, atol=1e-4)
    want_f, _ = OK.situ_and_mul(x, beta, linear_beta, False, total)
    got_f = situ_and_mul(x.cuda(), glc, glt, beta, linear_beta)
    assert got_f.dtype == dt and got_f.shape == (rows, cols // 2)
    assert torch.allclose(got_f.cpu()[:total].float(), want_f[:total].float(), **tol)
    if mode == "noquant":                      # d > 6144: the quantising entry point refuses, as the reference does
        with pytest.raises(NotImplementedError):
            situ_and_mul_quant(x.cuda(), glc, glt, beta, linear_beta)
        return
    with pytest.raises(NotImplementedError):
        situ_and_mul_quant(x.cuda(), glc, glt, beta, linear_beta, need_quant=False)
    with pytest.raises(NotImplementedError):
        situ_and_mul_quant(x.cuda(), glc, glt, beta, linear_beta, quant_type=1)
    want, ws = OK.situ_and_mul(x, beta, linear_beta, True, total)
    got, gs = situ_and_mul_quant(x.cuda(), glc, glt, beta, linear_beta)
    assert got.dtype == torch.int8 and gs.shape == (rows,)
    assert torch.allclose(gs.cpu()[:total], ws[:total], rtol=1e-5)
    d = (got.cpu()[:total].int() - want[:total].int()).abs()
    assert d.max() <= 1 and (d != 0).float().mean() < 2e-2
    if gl is not None:                         # the grouped entry point: same kernel, scale None without quantisation
        g2, s2 = situ(x.cuda(), glc, glt, need_quant=True, beta=beta, linear_beta=linear_beta)
        assert torch.equal(g2[:total], got[:total]) and torch.equal(s2[:total], gs[:total])
        g3, s3 = situ(x.cuda(), glc, glt, need_quant=False, beta=beta, linear_beta=linear_beta)
        assert s3 is None and torch.equal(g3[:total], got_f[:total])
        with pytest.raises(ValueError):
            situ(x.cuda(), glc, 2, need_quant=True)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,blocks,B,H,wdt", [(130, 8, 8, 7168, torch.float32), (17, 6, 3, 2048, None), (5, 4, 0, 64, None), (3, 70, 63, 8192, torch.float32)])
def test_attn_residual_mix(T, blocks, B, H, wdt, dt):
    """kimi_k3/attn_residual.py::mix_fused against the fp32 restatement: scores, softmax and mix in fp32, one rounding at the end -- within one
    step of the I/O dtype relative to the row's largest value.  Includes no bank rows at all (the output is the prefix row) and a strided bank."""
    from sgl_kernel_npu.kimi_k3.attn_residual import mix_fused
    torch.manual_seed(T + H)
    prefix = torch.randn(T, H).to(dt)
    bank_full = torch.randn(T, blocks + 1, H).to(dt)
    bank = bank_full[:, 1:]                                               # token stride (blocks + 1) * H: not the dense layout
    cw = (torch.randn(H) * 0.05).to(wdt or dt)
    want = OK.attn_residual_mix(prefix, bank, B, cw, 1e-6)
    got = mix_fused(prefix.cuda(), bank_full.cuda()[:, 1:], B, cw.cuda(), 1e-6)
    tol = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    err = (got.cpu().float() - want.float()).abs()
    assert bool((err <= tol * want.float().abs().amax(dim=-1, keepdim=True) + 1e-6).all()), float(err.max())
    if B == 0:
        assert torch.equal(got.cpu(), prefix)
    with pytest.raises(ValueError):
        mix_fused(prefix.cuda(), bank.cuda(), blocks + 1, cw.cuda(), 1e-6)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(4096, 7168), (3, 7), (129, 2880)])
def test_mul_add(rows, cols, dt):
    """moe/mul_add.py: routed * factor + shared in the tensors' dtype, bit for bit against the same expression restated in fp32 with the
    product's rounding made explicit."""
    from sgl_kernel_npu.moe.mul_add import mul_add
    torch.manual_seed(rows + cols)
    a, b = (torch.randn(rows, cols) * 2).to(dt), torch.randn(rows, cols).to(dt)
    got = mul_add(a.cuda(), b.cuda(), 2.5)
    assert got.dtype == dt and torch.equal(got.cpu(), OK.mul_add(a, b, 2.5))
    assert torch.equal(got.cpu(), a * 2.5 + b)                 # ... which is what torch's own evaluation of the expression gives


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("idx_dt", [torch.int32, torch.int64])
@pytest.mark.parametrize("S,K,D,E,Z", [(257, 8, 7168, 256, 64), (5, 12, 100, 16, 16), (64, 2, 512, 4, 1)])
def test_zero_experts_compute_identity(S, K, D, E, Z, idx_dt, dt):
    """moe/zero_experts_compute_identity.py: E real experts + Z zero experts; the result, the cleared scales and the rewritten indices (first
    one 0 when every selection of a token was a zero expert) against the restated kernel.  Integer effects exactly, the result within one
    rounding of the I/O dtype (the K scales are summed in another order)."""
    from sgl_kernel_npu.moe.zero_experts_compute_identity import zero_experts_compute_identity_triton
    torch.manual_seed(S + K + D)
    idx = torch.randint(0, E + Z, (S, K)).to(idx_dt)
    idx[0] = torch.arange(E, E + K) % (E + Z) if Z >= K else E          # a token whose selections are all zero experts
    idx[0] = torch.clamp(idx[0], min=E)
    if S > 1:
        idx[1] = torch.arange(K) % E                                    # ... and one with none
    scales = torch.rand(S, K)
    hidden = torch.randn(S, D).to(dt)
    want, widx, wsc = OK.zero_experts_compute_identity(idx, scales, E, hidden, identity_mask_value=7)
    gidx, gsc = idx.cuda(), scales.cuda()
    got = zero_experts_compute_identity_triton(gidx, gsc, E, "identity", hidden.cuda(), identity_mask_value=7)
    assert torch.equal(gidx.cpu(), widx) and torch.equal(gsc.cpu(), wsc)
    assert int(widx[0, 0]) == 0 and (widx[0, 1:] == 7).all()
    tol = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    assert torch.allclose(got.cpu().float(), want.float(), rtol=tol, atol=1e-6)
    assert (got.cpu()[1] == 0).all() if S > 1 else True


def test_split_qkv_rmsnorm_rope_pos_cache_half_replays_in_a_captured_graph():
    """The reference test replays the op in a captured device graph with new inputs in the same buffers
    (test_split_qkv_rmsnorm_rope_pos_cache_half_npu.py:213-260): positions are clamped inside the kernel, nothing synchronises."""
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope_pos_cache_half_npu import split_qkv_rmsnorm_rope_pos_cache_half_npu
    torch.manual_seed(1)
    B, qh, kvh, hd, max_pos = 12, 2048, 512, 128, 2048
    dt = torch.bfloat16
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = torch.einsum("i,j -> ij", torch.arange(max_pos, dtype=torch.float32), inv_freq)
    cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).cuda()
    qw, kw = torch.randn(hd).to(dt).cuda(), torch.randn(hd).to(dt).cuda()
    qkv_s = torch.randn(B, qh + 2 * kvh).to(dt).cuda()
    pos_s = torch.randint(0, max_pos, (B,), dtype=torch.int64).cuda()
    run = lambda: split_qkv_rmsnorm_rope_pos_cache_half_npu(qkv_s, pos_s, cache, qh, kvh, hd, eps=1e-6, q_weight=qw, k_weight=kw, rope_dim=hd)
    s_ = torch.cuda.Stream()
    s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        run()
    torch.cuda.current_stream().wait_stream(s_)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = run()
    for it in range(2):
        qkv_s.copy_(torch.randn(B, qh + 2 * kvh).to(dt))
        pos_s.copy_(torch.randint(0, max_pos, (B,), dtype=torch.int64))
        g.replay()
        torch.cuda.synchronize()
        want = OK.split_qkv_rmsnorm_rope_pos_cache_half(qkv_s.cpu(), pos_s.cpu(), cache.cpu(), qh, kvh, hd, 1e-6, qw.cpu(), kw.cpu(), None, None, hd)
        assert torch.equal(outs[2].cpu(), want[2])
        assert torch.allclose(outs[0].cpu().float(), want[0].float(), rtol=2 ** -7, atol=2e-3)
        assert torch.allclose(outs[1].cpu().float(), want[1].float(), rtol=2 ** -7, atol=2e-3)


def _mla_pre_inputs(N, Hq, hidden, dt=torch.bfloat16):
    torch.manual_seed(42)
    d = dict(hid=(torch.randn(N, hidden) * 0.5).to(dt), wdqkv=torch.randint(-8, 8, (2112, hidden), dtype=torch.int8),
             wuq=torch.randint(-8, 8, (Hq * 192, 1536), dtype=torch.int8), descale0=(torch.rand(2112) * 1e-3 + 5e-4).float(),
             descale1=(torch.rand(Hq * 192) * 1e-3 + 5e-4).float(), bias0=torch.randint(-50, 50, (2112,), dtype=torch.int32),
             bias1=torch.randint(-50, 50, (Hq * 192,), dtype=torch.int32), gamma0=torch.randn(hidden).to(dt), beta0=torch.randn(hidden).to(dt),
             gamma1=torch.randn(1536).to(dt), beta1=(torch.randn(1536) * 0.1).to(dt), gamma2=torch.randn(512).to(dt),
             wuk=(torch.randn(Hq, 128, 512) * 0.1).to(dt), cos=torch.rand(N, 64).to(dt), sin=torch.rand(N, 64).to(dt),
             qs0=torch.tensor([0.02]).to(dt), qo0=torch.tensor([3], dtype=torch.int8), qs1=torch.tensor([0.03]).to(dt),
             qo1=torch.tensor([-2], dtype=torch.int8))
    return d


def _mla_pre_exact(z, eps=1e-6):
    """The same network in float64 with the golden's rounding / quantisation points kept (what both the kernel and the fp32
    oracle approximate): exact statistics for the two RMSNorms, exact BMM accumulation."""
    dt = z["hid"].dtype
    q8 = OK._quant_per_tensor(z["hid"], z["qs0"], z["qo0"])
    f = OK._int8_gemm_dequant(q8, z["wdqkv"], z["descale0"], z["bias0"], dt).double()
    k_nope, k_pe, q = f[:, :512], f[:, 512:576], f[:, 576:]
    rms = lambda x, g: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * g.double()
    qn = (rms(q, z["gamma1"]) + z["beta1"].double()).float()
    y = OK._int8_gemm_dequant(OK._quant_per_tensor(qn, z["qs1"], z["qo1"]), z["wuq"], z["descale1"], z["bias1"], dt)
    y = y.view(y.shape[0], -1, 192).double()
    c, s_ = z["cos"].double().unsqueeze(1), z["sin"].double().unsqueeze(1)
    rot = lambda t: torch.cat([-t[..., 32:], t[..., :32]], -1)
    q0 = torch.einsum("nhk,hkd->nhd", y[..., :128], z["wuk"].double())
    q1 = y[..., 128:] * c + rot(y[..., 128:]) * s_
    kp = k_pe.unsqueeze(1)
    return q0, q1, rms(k_nope, z["gamma2"]), (kp * c + rot(kp) * s_).squeeze(1)


_MLA_PRE_MISS = {}          # measured fraction of elements outside the reference's atol = rtol = 1e-3, per (case, output)


def _record_mla_pre_miss(key, frac):
    """Keeps the measured miss fractions of a test session and mirrors them to gpurun_out/mla_pre_miss_fraction.json (the copy under
    profiles/ is what DESIGN.md quotes); never fails the test."""
    import json
    import os
    _MLA_PRE_MISS[key] = frac
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "mla_pre_miss_fraction.json"), "w") as f:
            json.dump(_MLA_PRE_MISS, f, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.mark.parametrize("N,Hq,hidden,dt", [(1, 32, 7168, torch.bfloat16), (16, 64, 7168, torch.bfloat16), (31, 128, 7168, torch.bfloat16),
                                            (31, 128, 6144, torch.bfloat16), (70, 16, 2048, torch.bfloat16), (128, 128, 7168, torch.bfloat16),
                                            (1024, 16, 7168, torch.bfloat16), (200, 128, 7168, torch.bfloat16),
                                            # fp16 end to end (reference mla_preprocess_mix_fp16.hpp): same chain, half-precision I/O
                                            (31, 128, 7168, torch.float16), (70, 16, 2048, torch.float16), (128, 128, 7168, torch.float16),
                                            (200, 32, 6144, torch.float16)],
                         ids=lambda v: str(v).replace("torch.", "") if isinstance(v, torch.dtype) else str(v))
def test_mla_preprocess(N, Hq, hidden, dt):
    """torch.ops.npu.mla_preprocess vs the transcription of the reference golden (seed 42, shapes of
    tests/python/sgl_kernel_npu/test_mla_preprocess.py:487-498, plus the decode batch 128, the maximum 1024 tokens and a token
    count that is not a multiple of the 128-row block).

    Bar.  The reference asserts atol = rtol = 1e-3 against golden2_pytorch (:700-745).  Kernel and golden run the same chain of
    exact stages (int8 GEMMs, element-wise fp32 arithmetic with identical rounding points) except for the ORDER of the fp32
    sums inside the two RMSNorms and the BMM; a last-bit difference there moves a bf16 rounding (or, before GEMM2, an int8
    rounding) in a handful of elements.  So: (1) all but a vanishing fraction of the elements meet the reference's 1e-3,
    (2) the rest are bounded by what one flipped int8 step can do, (3) measured against the float64 evaluation of the same
    network, the kernel is as accurate as the golden (mean absolute error, which a few flips do not dominate).
    The measured fraction of (1) per output is recorded (profiles/r04_mla_pre_miss_fraction.json) and the bar is set from it."""
    block_size, nblocks = 128, max(4, (N + 127) // 128 + 1)
    z = _mla_pre_inputs(N, Hq, hidden, dt)
    slots = torch.randperm(nblocks * block_size)[:N].to(torch.int32)
    want = OK.mla_preprocess(z["hid"], z["wdqkv"], z["descale0"], z["bias0"], z["gamma1"], z["beta1"], z["gamma2"], z["wuq"], z["descale1"],
                             z["bias1"], z["wuk"], z["cos"], z["sin"], z["qs0"], z["qo0"], z["qs1"], z["qo1"])
    d = lambda t: t.cuda()
    kv = torch.zeros((nblocks, block_size, 1, 512), dtype=dt, device="cuda")
    kr = torch.zeros((nblocks, block_size, 1, 64), dtype=dt, device="cuda")
    q0 = torch.empty((N, Hq, 512), dtype=dt, device="cuda")
    q1 = torch.empty((N, Hq, 64), dtype=dt, device="cuda")
    for _ in range(2):
        out = torch.ops.npu.mla_preprocess(d(z["hid"]), d(z["gamma0"]), d(z["beta0"]), d(z["wdqkv"]), d(z["descale0"]), d(z["gamma1"]),
                                           d(z["beta1"]), d(z["wuq"]), d(z["descale1"]), d(z["gamma2"]), d(z["cos"]), d(z["sin"]), d(z["wuk"]),
                                           kv, kr, d(slots), d(z["qs0"]), d(z["qo0"]), d(z["bias0"]), d(z["qs1"]), d(z["qo1"]), d(z["bias1"]),
                                           cache_mode="krope_ctkv", quant_mode="per_tensor_quant_asymm", q_out0=q0, kv_cache_out0=kv,
                                           q_out1=q1, kv_cache_out1=kr)
    assert out[0].data_ptr() == q0.data_ptr() and out[1].data_ptr() == kv.data_ptr()
    k_nope = kv.view(-1, 512)[slots.long().cuda()].cpu()
    k_pe = kr.view(-1, 64)[slots.long().cuda()].cpu()
    exact = _mla_pre_exact(z)
    got = (q0.cpu(), q1.cpu(), k_nope, k_pe)
    for name, g, w, ex in zip(("q_out0", "q_out1", "k_nope", "k_pe"), got, want, exact):
        g64, w64 = g.double(), w.double()
        bad = ~torch.isclose(g64, w64, rtol=1e-3, atol=1e-3)
        _record_mla_pre_miss(f"{N}x{Hq}x{hidden}_{str(dt).replace('torch.', '')}_{name}", bad.double().mean().item())
        # (1) measured in round 4 (profiles/r04_mla_pre_miss_fraction.json, every case of this test): k_nope and k_pe meet the reference's
        # bar on EVERY element; q_out0 / q_out1 miss it on <= 1.3e-5 of the elements except in the 1024-token case (1.3e-3 / 7.2e-4: one
        # int8 step flipped in front of GEMM2 moves a whole output row of a head).  The bar follows the measurement with a margin.
        bar = 0.0 if name in ("k_nope", "k_pe") else (2e-3 if N >= 512 else 1e-4)
        assert bad.double().mean().item() <= bar, (name, bad.double().mean().item(), bar)
        assert torch.allclose(g64, w64, rtol=2 ** -5, atol=5e-2), (name, (g64 - w64).abs().max().item())  # (2)
        err_k, err_o = (g64 - ex).abs().mean().item(), (w64 - ex).abs().mean().item()
        assert err_k <= 1.05 * err_o + 1e-7, (name, err_k, err_o)                                       # (3)
    assert torch.equal(k_pe, want[3]), "k_pe has no reduction in it: it must match the golden bit for bit"
    # untouched cache rows stay zero
    mask = torch.ones(nblocks * block_size, dtype=torch.bool)
    mask[slots.long()] = False
    assert not kv.view(-1, 512).cpu()[mask].any()


@pytest.mark.parametrize("N,Hq,hidden,dt,qmode,cmode", [
    (128, 128, 7168, torch.bfloat16, "per_tensor_quant_asymm", "krope_ctkv"),       # the decode batch: 256 workgroups, one per CU
    (128, 128, 7168, torch.bfloat16, "per_token_quant_symm", "krope_ctkv"),
    (31, 128, 7168, torch.float16, "per_tensor_quant_asymm", "nzcache"),
    (64, 32, 6144, torch.bfloat16, "per_token_quant_symm", "int8_nzcache"),
    (1, 16, 2048, torch.bfloat16, "per_tensor_quant_asymm", "krope_ctkv"),
    (200, 32, 7168, torch.bfloat16, "per_tensor_quant_asymm", "krope_ctkv"),        # two token blocks per head
])
def test_mla_preprocess_one_launch_equals_four_launches(N, Hq, hidden, dt, qmode, cmode):
    """The op as ONE launch (stage bodies behind grid barriers, the default at decode sizes) against the four launches
    (MI_MLA_PRE_ONE_LAUNCH=0): every output and both caches bit for bit, repeatedly on the same buffers (the barrier words' epoch moves on)."""
    import os
    block_size, nblocks = 128, max(4, (N + 127) // 128 + 1)
    z = _mla_pre_inputs(N, Hq, hidden, dt)
    slots = torch.randperm(nblocks * block_size)[:N].to(torch.int32)
    d = lambda t: t.cuda()
    int8c = cmode == "int8_nzcache"
    extra = dict(ctkv_scale=torch.tensor([0.07]).to(dt).cuda(), q_nope_scale=(torch.rand(Hq) * 0.5 + 0.5).to(dt).cuda()) if int8c else {}

    def run():
        kv = torch.zeros((nblocks, block_size, 1, 512), dtype=torch.int8 if int8c else dt, device="cuda")
        kr = torch.zeros((nblocks, block_size, 1, 64), dtype=dt, device="cuda")
        q0 = torch.empty((N, Hq, 512), dtype=torch.int8 if int8c else dt, device="cuda")
        q1 = torch.empty((N, Hq, 64), dtype=dt, device="cuda")
        torch.ops.npu.mla_preprocess(d(z["hid"]), d(z["gamma0"]), d(z["beta0"]), d(z["wdqkv"]), d(z["descale0"]), d(z["gamma1"]), d(z["beta1"]),
                                     d(z["wuq"]), d(z["descale1"]), d(z["gamma2"]), d(z["cos"]), d(z["sin"]), d(z["wuk"]), kv, kr, d(slots),
                                     d(z["qs0"]), d(z["qo0"]), d(z["bias0"]), d(z["qs1"]), d(z["qo1"]), d(z["bias1"]), cache_mode=cmode,
                                     quant_mode=qmode, q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr, **extra)
        torch.cuda.synchronize()
        return q0, q1, kv, kr

    old = os.environ.get("MI_MLA_PRE_ONE_LAUNCH")
    try:
        os.environ["MI_MLA_PRE_ONE_LAUNCH"] = "0"
        want = run()
        os.environ["MI_MLA_PRE_ONE_LAUNCH"] = "1"
        for rep in range(3):
            got = run()
            for name, g, w in zip(("q_out0", "q_out1", "kv_cache", "kv_cache_rope"), got, want):
                assert torch.equal(g, w), (name, rep)
    finally:
        if old is None:
            os.environ.pop("MI_MLA_PRE_ONE_LAUNCH", None)
        else:
            os.environ["MI_MLA_PRE_ONE_LAUNCH"] = old


def test_mla_preprocess_under_inference_mode_and_foreign_wuk_dtype():
    """Inference tensors carry no version counter (at::Tensor::_version() throws): the wuk re-layout cache must not ask for one.
    And a wuk stored in another dtype than the activations is converted inside the cache's make step, keyed on the caller's tensor:
    same bytes as the plain call, twice (second call = cache hit)."""
    dt = torch.bfloat16
    N, Hq, hidden, block_size, nblocks = 17, 16, 2048, 128, 4
    z = _mla_pre_inputs(N, Hq, hidden, dt)
    slots = torch.randperm(nblocks * block_size)[:N].to(torch.int32)

    def run(wuk_dtype):
        d = lambda t: t.cuda()
        kv = torch.zeros((nblocks, block_size, 1, 512), dtype=dt, device="cuda")
        kr = torch.zeros((nblocks, block_size, 1, 64), dtype=dt, device="cuda")
        q0 = torch.empty((N, Hq, 512), dtype=dt, device="cuda")
        q1 = torch.empty((N, Hq, 64), dtype=dt, device="cuda")
        wuk = d(z["wuk"]).to(wuk_dtype)
        outs = []
        for _ in range(2):
            torch.ops.npu.mla_preprocess(d(z["hid"]), d(z["gamma0"]), d(z["beta0"]), d(z["wdqkv"]), d(z["descale0"]), d(z["gamma1"]),
                                         d(z["beta1"]), d(z["wuq"]), d(z["descale1"]), d(z["gamma2"]), d(z["cos"]), d(z["sin"]), wuk,
                                         kv, kr, d(slots), d(z["qs0"]), d(z["qo0"]), d(z["bias0"]), d(z["qs1"]), d(z["qo1"]), d(z["bias1"]),
                                         cache_mode="krope_ctkv", quant_mode="per_tensor_quant_asymm", q_out0=q0, kv_cache_out0=kv,
                                         q_out1=q1, kv_cache_out1=kr)
            outs.append([t.clone() for t in (q0, q1, kv, kr)])
        assert all(torch.equal(a, b) for a, b in zip(*outs))
        return outs[0]

    plain = run(dt)
    with torch.inference_mode():
        inf = run(dt)
        inf32 = run(torch.float32)              # bf16 values held in fp32: converts back exactly
    for a, b, c in zip(plain, inf, inf32):
        assert torch.equal(a, b) and torch.equal(a, c)


def _mla_pre_exact_token(z, eps=1e-6):
    """float64 evaluation of the per-token network with the kernel's rounding / quantisation points kept."""
    dt = z["hid"].dtype
    a8, t0 = OK._quant_per_token(z["hid"])
    f = OK._int8_gemm_dequant_token(a8, z["wdqkv"], z["descale0"], t0, dt).double()
    k_nope, k_pe, q = f[:, :512], f[:, 512:576], f[:, 576:]
    rms = lambda x, g: x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * g.double()
    qn = (rms(q, z["gamma1"]) + z["beta1"].double()).float()
    q8, t1 = OK._quant_per_token(qn)
    y = OK._int8_gemm_dequant_token(q8, z["wuq"], z["descale1"], t1, dt)
    y = y.view(y.shape[0], -1, 192).double()
    c, s_ = z["cos"].double().unsqueeze(1), z["sin"].double().unsqueeze(1)
    rot = lambda t: torch.cat([-t[..., 32:], t[..., :32]], -1)
    kp = k_pe.unsqueeze(1)
    return (torch.einsum("nhk,hkd->nhd", y[..., :128], z["wuk"].double()), y[..., 128:] * c + rot(y[..., 128:]) * s_,
            rms(k_nope, z["gamma2"]), (kp * c + rot(kp) * s_).squeeze(1))


@pytest.mark.parametrize("N,Hq,hidden", [(1, 32, 7168), (31, 128, 7168), (70, 16, 2048), (128, 128, 7168), (300, 16, 6144)])
def test_mla_preprocess_per_token_quant(N, Hq, hidden):
    """quant_mode='per_token_quant_symm', the reference's default.  The reference tests hold no golden for this mode, so the
    oracle restates the kernel (oracle/kernels.py mla_preprocess_per_token: PARITY UNPINNED); same three-part bar as
    test_mla_preprocess, plus the sanity that per-token quantisation is at least as accurate as per-tensor against the
    unquantised float64 network would be out of scope here."""
    dt = torch.bfloat16
    block_size, nblocks = 128, max(4, (N + 127) // 128 + 1)
    z = _mla_pre_inputs(N, Hq, hidden, dt)
    z["hid"][0, :] = 0 if N > 2 else z["hid"][0, :]               # an all-zero token: scale 0, zeros out
    slots = torch.randperm(nblocks * block_size)[:N].to(torch.int32)
    want = OK.mla_preprocess_per_token(z["hid"], z["wdqkv"], z["descale0"], z["gamma1"], z["beta1"], z["gamma2"], z["wuq"], z["descale1"],
                                       z["wuk"], z["cos"], z["sin"])
    d = lambda t: t.cuda()
    kv = torch.zeros((nblocks, block_size, 1, 512), dtype=dt, device="cuda")
    kr = torch.zeros((nblocks, block_size, 1, 64), dtype=dt, device="cuda")
    q0 = torch.empty((N, Hq, 512), dtype=dt, device="cuda")
    q1 = torch.empty((N, Hq, 64), dtype=dt, device="cuda")
    torch.ops.npu.mla_preprocess(d(z["hid"]), d(z["gamma0"]), d(z["beta0"]), d(z["wdqkv"]), d(z["descale0"]), d(z["gamma1"]), d(z["beta1"]),
                                 d(z["wuq"]), d(z["descale1"]), d(z["gamma2"]), d(z["cos"]), d(z["sin"]), d(z["wuk"]), kv, kr, d(slots),
                                 d(z["qs0"]), d(z["qo0"]), d(z["bias0"]), d(z["qs1"]), d(z["qo1"]), d(z["bias1"]), cache_mode="krope_ctkv",
                                 quant_mode="per_token_quant_symm", q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
    k_nope = kv.view(-1, 512)[slots.long().cuda()].cpu()
    k_pe = kr.view(-1, 64)[slots.long().cuda()].cpu()
    exact = _mla_pre_exact_token(z)
    for name, g, w, ex in zip(("q_out0", "q_out1", "k_nope", "k_pe"), (q0.cpu(), q1.cpu(), k_nope, k_pe), want, exact):
        assert torch.isfinite(g.float()).all(), name
        g64, w64 = g.double(), w.double()
        bad = ~torch.isclose(g64, w64, rtol=1e-3, atol=1e-3)
        assert bad.double().mean().item() <= 2e-3, (name, bad.double().mean().item())
        assert torch.allclose(g64, w64, rtol=2 ** -5, atol=5e-2), (name, (g64 - w64).abs().max().item())
        err_k, err_o = (g64 - ex).abs().mean().item(), (w64 - ex).abs().mean().item()
        assert err_k <= 1.05 * err_o + 1e-7, (name, err_k, err_o)
    assert torch.equal(k_pe, want[3])


@pytest.mark.parametrize("cache_mode", ["nzcache", "int8_nzcache"])
@pytest.mark.parametrize("N,Hq,hidden,block_size", [(1, 32, 7168, 128), (31, 128, 7168, 128), (128, 128, 7168, 128), (70, 16, 2048, 64)])
def test_mla_preprocess_nz_cache_modes(cache_mode, N, Hq, hidden, block_size):
    """cache_mode 'nzcache' / 'int8_nzcache' (csrc/mla_preprocess/op_host/mla_preprocess.cpp:605-606; the reference test runs all three
    modes, tests/python/sgl_kernel_npu/test_mla_preprocess.py:504-600).  Caches are read back the way the reference test does
    (extract_from_nzcache with C0 = 16, or 32 for the int8 k_nope), and nothing outside the written slots may change.
    nzcache: the VALUES must be bit-identical to what krope_ctkv writes for the same inputs (only the layout differs).
    int8_nzcache: against the golden's int8 outputs -- an int8 step is what one flipped rounding upstream can do, so: equal up to
    +-1, and all but a vanishing fraction equal exactly."""
    dt = torch.bfloat16
    nblocks = max(3, (N + block_size - 1) // block_size + 1)
    z = _mla_pre_inputs(N, Hq, hidden, dt)
    torch.manual_seed(7)
    slots = torch.randperm(nblocks * block_size)[:N].to(torch.int32)
    ctkv_scale = torch.tensor([0.37]).to(dt)                      # reference test: uniform(-2, 2) / uniform(-1, 1); avoid ~0 here
    qnope_scale = (torch.rand(Hq) * 1.5 + 0.25).to(dt) * torch.where(torch.rand(Hq) < 0.5, -1.0, 1.0).to(dt)
    int8 = cache_mode == "int8_nzcache"
    d = lambda t: t.cuda()

    def run(mode, kv, kr, q0, q1):
        kw = dict(ctkv_scale=d(ctkv_scale), q_nope_scale=d(qnope_scale)) if mode == "int8_nzcache" else {}
        torch.ops.npu.mla_preprocess(d(z["hid"]), d(z["gamma0"]), d(z["beta0"]), d(z["wdqkv"]), d(z["descale0"]), d(z["gamma1"]),
                                     d(z["beta1"]), d(z["wuq"]), d(z["descale1"]), d(z["gamma2"]), d(z["cos"]), d(z["sin"]), d(z["wuk"]),
                                     kv, kr, d(slots), d(z["qs0"]), d(z["qo0"]), d(z["bias0"]), d(z["qs1"]), d(z["qo1"]), d(z["bias1"]),
                                     cache_mode=mode, quant_mode="per_tensor_quant_asymm", q_out0=q0, kv_cache_out0=kv, q_out1=q1,
                                     kv_cache_out1=kr, **kw)

    mk = lambda dim, dtype: torch.zeros((nblocks, block_size, 1, dim), dtype=dtype, device="cuda")
    kv, kr = mk(512, torch.int8 if int8 else dt), mk(64, dt)
    q0 = torch.empty((N, Hq, 512), dtype=torch.int8 if int8 else dt, device="cuda")
    q1 = torch.empty((N, Hq, 64), dtype=dt, device="cuda")
    run(cache_mode, kv, kr, q0, q1)
    kv_ref, kr_ref = mk(512, dt), mk(64, dt)
    q0_ref, q1_ref = torch.empty((N, Hq, 512), dtype=dt, device="cuda"), torch.empty((N, Hq, 64), dtype=dt, device="cuda")
    run("krope_ctkv", kv_ref, kr_ref, q0_ref, q1_ref)
    kv_h, kr_h = kv.cpu(), kr.cpu()
    k_nope = torch.stack([OK.extract_from_nzcache(kv_h, s_, 32 if int8 else 16) for s_ in slots.tolist()])
    k_pe = torch.stack([OK.extract_from_nzcache(kr_h, s_, 16) for s_ in slots.tolist()])
    # the rope cache and q_out1 do not depend on the int8 path: bit-identical to the krope_ctkv run, just laid out differently
    assert torch.equal(k_pe, kr_ref.view(-1, 64)[slots.long().cuda()].cpu())
    assert torch.equal(q1.cpu(), q1_ref.cpu())
    if not int8:
        assert torch.equal(k_nope, kv_ref.view(-1, 512)[slots.long().cuda()].cpu())
        assert torch.equal(q0.cpu(), q0_ref.cpu())
    else:
        want = OK.mla_preprocess(z["hid"], z["wdqkv"], z["descale0"], z["bias0"], z["gamma1"], z["beta1"], z["gamma2"], z["wuq"],
                                 z["descale1"], z["bias1"], z["wuk"], z["cos"], z["sin"], z["qs0"], z["qo0"], z["qs1"], z["qo1"],
                                 cache_mode="int8_nzcache", ctkv_scale=ctkv_scale, qnope_scale=qnope_scale)
        for name, g, w in (("q_out0", q0.cpu(), want[0]), ("k_nope", k_nope, want[2])):
            diff = (g.int() - w.int()).abs()
            assert diff.max().item() <= 1, (name, diff.max().item())
            assert (diff != 0).double().mean().item() <= 5e-3, (name, (diff != 0).double().mean().item())
        # the int8 q is exactly the golden's quantisation of the value the bf16 path writes (same kernel, one more rounding step)
        q0_from_bf16 = OK._quant_per_tensor_muls(q0_ref.cpu(), qnope_scale.reshape(1, Hq, 1), torch.zeros(1))
        assert torch.equal(q0.cpu(), q0_from_bf16)
    # slots that were not written stay zero: count the non-zero elements per cache
    written = set(slots.tolist())
    for cache, dim, c0 in ((kv_h, 512, 32 if int8 else 16), (kr_h, 64, 16)):
        keep = torch.ones(cache.numel(), dtype=torch.bool)
        for s_ in written:
            keep[OK.nz_cache_offsets(s_, block_size, dim, c0)] = False
        assert not cache.reshape(-1)[keep].any()


def test_mla_preprocess_rejects_modes_it_does_not_implement():
    """Unknown cache / quant modes and inconsistent int8_nzcache arguments must fail loudly; an omitted quant_mode selects
    per_token_quant_symm, as in the reference (csrc/mla_preprocess/op_host/mla_preprocess.cpp:634-635)."""
    dt = torch.bfloat16
    z = _mla_pre_inputs(2, 16, 2048, dt)
    d = lambda t: t.cuda()
    kv = torch.zeros((2, 128, 1, 512), dtype=dt, device="cuda")
    kr = torch.zeros((2, 128, 1, 64), dtype=dt, device="cuda")
    q0, q1 = torch.empty((2, 16, 512), dtype=dt, device="cuda"), torch.empty((2, 16, 64), dtype=dt, device="cuda")
    slots = torch.tensor([0, 5], dtype=torch.int32, device="cuda")
    args = (d(z["hid"]), d(z["gamma0"]), d(z["beta0"]), d(z["wdqkv"]), d(z["descale0"]), d(z["gamma1"]), d(z["beta1"]), d(z["wuq"]),
            d(z["descale1"]), d(z["gamma2"]), d(z["cos"]), d(z["sin"]), d(z["wuk"]), kv, kr, slots, d(z["qs0"]), d(z["qo0"]), d(z["bias0"]),
            d(z["qs1"]), d(z["qo1"]), d(z["bias1"]))
    outs = dict(q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
    # int8_nzcache without its scales / with bf16 outputs is inconsistent; "nz" is not a mode
    for bad in (dict(cache_mode="krope_ctkv", quant_mode="per_channel"), dict(cache_mode="int8_nzcache", quant_mode="per_tensor_quant_asymm"),
                dict(cache_mode="nz", quant_mode="per_tensor_quant_asymm")):
        with pytest.raises(RuntimeError):
            torch.ops.npu.mla_preprocess(*args, **bad, **outs)
    torch.ops.npu.mla_preprocess(*args, cache_mode="krope_ctkv", quant_mode="per_tensor_quant_asymm", **outs)      # the built modes run
    a = torch.ops.npu.mla_preprocess(*args, quant_mode="per_token_quant_symm", **outs)[0].clone()
    b = torch.ops.npu.mla_preprocess(*args, **outs)[0]              # omitted quant_mode == the reference's default, per-token
    assert torch.equal(a, b)


def _sgl_lib():
    import ctypes
    from capi import load
    L = load("libmi_sgl_kernels.so")
    V, I = ctypes.c_void_p, ctypes.c_int
    L.mi_mla_pre_gemm_i8.argtypes = [V, I, I, V, I, I, V, V, V, V, V, I, V]
    L.mi_mla_pre_bmm_rope.argtypes = [V, I, I, V, V, V, I, V, V, V, V]
    L.mi_mla_pre_gemm_i8_partials.argtypes = [I]
    L.mi_mla_pre_gemm2_bmm_rope.argtypes = [V, I, V, I, V, V, V, V, V, V, I, V, V, V, V]
    L.mi_mla_pre_gemm2_bmm_rope.restype = I
    L.mi_mla_pre_gemm_i8.restype = L.mi_mla_pre_bmm_rope.restype = L.mi_mla_pre_gemm_i8_partials.restype = I
    return L


@pytest.mark.parametrize("M,K,N", [(1, 7168, 2112), (128, 7168, 2112), (130, 6144, 2112), (77, 2048, 2112), (128, 1536, 24576),
                                   (1024, 1536, 3072), (5, 192, 100), (128, 1536, 16384), (40, 1536, 8192 + 48), (3, 1536, 50)])
def test_mla_pre_skinny_int8_gemm_exact(M, K, N):
    """mi_mla_pre_gemm_i8 through the C-ABI: split-K atomics (mode 0) give the exact int32 product; mode 1 = the golden's
    dequant (int32 + bias) * descale -> bf16 with one rounding.  Includes K that is not a multiple of the 512-byte chunk and
    column / row counts that are not multiples of the tiles."""
    from capi import ptr, stream_ptr
    L = _sgl_lib()
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    a = torch.randint(-128, 128, (M, K), generator=g, device="cuda", dtype=torch.int32).to(torch.int8)
    w = torch.randint(-128, 128, (N, K), generator=g, device="cuda", dtype=torch.int32).to(torch.int8)
    want = torch.round(a.double() @ w.double().T)                    # exact in float64
    parts = L.mi_mla_pre_gemm_i8_partials(K)
    assert parts == (K + 511) // 512
    c = torch.full((parts, M, N), 12345, dtype=torch.int32, device="cuda")      # no zero-fill needed: every slice is overwritten
    assert L.mi_mla_pre_gemm_i8(ptr(a), M, K, ptr(w), N, 0, ptr(c), None, None, None, None, 0, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(c.double().sum(0), want)
    bias = torch.randint(-50, 50, (N,), generator=g, device="cuda", dtype=torch.int32)
    descale = torch.rand(N, generator=g, device="cuda") * 1e-3 + 5e-4
    for dtype, code in ((torch.bfloat16, 0), (torch.float16, 1)):
        y = torch.zeros((M, N), dtype=dtype, device="cuda")
        assert L.mi_mla_pre_gemm_i8(ptr(a), M, K, ptr(w), N, 1, None, ptr(bias), ptr(descale), None, ptr(y), code, stream_ptr()) == 0
        torch.cuda.synchronize()
        y_want = ((want.to(torch.int32) + bias).float() * descale).to(dtype)
        assert torch.equal(y.view(torch.int16), y_want.view(torch.int16)), dtype
        # per-token form: no bias, (channel scale) then (token scale), one rounding
        rs = torch.rand(M, generator=g, device="cuda") * 0.05 + 0.01
        assert L.mi_mla_pre_gemm_i8(ptr(a), M, K, ptr(w), N, 1, None, None, ptr(descale), ptr(rs), ptr(y), code, stream_ptr()) == 0
        torch.cuda.synchronize()
        y_want = ((want.float() * descale) * rs[:, None]).to(dtype)
        assert torch.equal(y.view(torch.int16), y_want.view(torch.int16)), dtype


@pytest.mark.parametrize("M,Hq", [(1, 8), (128, 128), (33, 16), (300, 4)])
@pytest.mark.parametrize("dtype,code", [(torch.bfloat16, 0), (torch.float16, 1)])
def test_mla_pre_bmm_rope(M, Hq, dtype, code):
    from capi import ptr, stream_ptr
    L = _sgl_lib()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + Hq)
    y = torch.randn((M, Hq * 192), generator=g, device="cuda").to(dtype)
    wuk = (torch.randn((Hq, 128, 512), generator=g, device="cuda") * 0.1).to(dtype)
    cos, sin = torch.rand((M, 64), generator=g, device="cuda").to(dtype), torch.rand((M, 64), generator=g, device="cuda").to(dtype)
    q0 = torch.zeros((M, Hq, 512), dtype=dtype, device="cuda")
    q1 = torch.zeros((M, Hq, 64), dtype=dtype, device="cuda")
    wuk_t = wuk.transpose(1, 2).contiguous()
    assert L.mi_mla_pre_bmm_rope(ptr(y), M, Hq, ptr(wuk_t), ptr(cos), ptr(sin), code, ptr(q0), ptr(q1), None, stream_ptr()) == 0
    torch.cuda.synchronize()
    yv = y.view(M, Hq, 192)
    exact = torch.einsum("nhk,hkd->nhd", yv[..., :128].double(), wuk.double())
    # fp32 accumulation of exact products, one rounding: within one output ulp of the float64 result
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert bool(((q0.double() - exact).abs() <= ulp * exact.abs() + 1e-6).all())
    pe = yv[..., 128:].float()
    rot = torch.cat([-pe[..., 32:], pe[..., :32]], -1)
    want1 = (pe * cos.float().unsqueeze(1) + rot * sin.float().unsqueeze(1)).to(dtype)
    assert torch.equal(q1.view(torch.int16), want1.view(torch.int16))


@pytest.mark.parametrize("T,Hq,Hk,D,R", [(32, 8, 1, 64, 32), (32, 8, 1, 32, 32), (17, 16, 1, 128, 64), (64, 32, 1, 128, 64),
                                         (5, 128, 1, 192, 64),        # reference cases (test_fused_rope_qk_mqa.py:64-71) + MLA-sized
                                         (300, 128, 1, 192, 64),      # many workgroups of the 16-byte-lane kernel, heads straddling them
                                         (7, 4, 1, 40, 20)])          # R % 16 != 0: the scalar kernel
@pytest.mark.parametrize("neox", [True, False])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_rope_qk_mqa(T, Hq, Hk, D, R, neox, dtype):
    from sgl_kernel_npu.norm.fused_rope_qk_mqa import fused_rope_qk_mqa
    torch.manual_seed(3)
    q, k = torch.randn(T, Hq, D).to(dtype), torch.randn(T, Hk, D).to(dtype)
    cs = torch.randn(T, R).to(dtype)
    wq, wk = OK.fused_rope_qk_mqa(q, k, cs, R, neox)
    oq, ok = fused_rope_qk_mqa(q.cuda(), k.cuda(), cs.cuda(), R, neox)
    # op-by-op rounding in the I/O dtype is reproduced, so the result is bit-identical (reference: assert_close defaults)
    assert torch.equal(oq.cpu(), wq) and torch.equal(ok.cpu(), wk)
    # strided query (a slice of a wider tensor), as SGLang passes views
    wide = torch.randn(T, Hq, D + 16).to(dtype).cuda()
    oq2, _ = fused_rope_qk_mqa(wide[..., :D], k.cuda(), cs.cuda(), R, neox)
    assert torch.equal(oq2.cpu(), OK.fused_rope_qk_mqa(wide[..., :D].cpu(), k, cs, R, neox)[0])


@pytest.mark.parametrize("M,Hq,per_token,dt", [(128, 128, False, torch.bfloat16), (1, 16, False, torch.bfloat16), (70, 32, True, torch.bfloat16),
                                              (200, 16, False, torch.float16), (31, 128, True, torch.bfloat16),
                                              # 128-row workgroups (heads x row blocks >= 256); the cases above take the 64-row form
                                              (300, 128, False, torch.bfloat16), (257, 128, True, torch.float16)])
def test_mla_pre_fused_gemm2_bmm_rope_equals_two_launches(M, Hq, per_token, dt):
    """mi_mla_pre_gemm2_bmm_rope (one launch, the GEMM2 output stays in LDS) against mi_mla_pre_gemm_i8(mode 1) + mi_mla_pre_bmm_rope:
    same MFMA shapes and accumulation order, so q_out0 and q_out1 must match BIT FOR BIT -- which carries every parity statement of
    the two-launch kernels (exact int32 products, golden rounding points) over to the fused one."""
    from capi import ptr, stream_ptr
    L = _sgl_lib()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + Hq)
    code = 0 if dt == torch.bfloat16 else 1                       # MI_DTYPE_BF16 / MI_DTYPE_F16
    a8 = torch.randint(-127, 128, (M, 1536), generator=g, device="cuda", dtype=torch.int8)
    wuq = torch.randint(-8, 8, (Hq * 192, 1536), generator=g, device="cuda", dtype=torch.int8)
    descale = torch.rand(Hq * 192, generator=g, device="cuda") * 1e-3 + 5e-4
    bias = None if per_token else torch.randint(-50, 50, (Hq * 192,), generator=g, device="cuda", dtype=torch.int32)
    rscale = (torch.rand(M, generator=g, device="cuda") * 0.02 + 0.01) if per_token else None
    wuk_t = (torch.randn((Hq, 512, 128), generator=g, device="cuda") * 0.1).to(dt)
    cos, sin = torch.rand((M, 64), generator=g, device="cuda").to(dt), torch.rand((M, 64), generator=g, device="cuda").to(dt)
    y = torch.empty((M, Hq * 192), dtype=dt, device="cuda")
    q0a, q1a = torch.full((M, Hq, 512), 7.0, dtype=dt, device="cuda"), torch.full((M, Hq, 64), 7.0, dtype=dt, device="cuda")
    q0b, q1b = torch.full_like(q0a, 9.0), torch.full_like(q1a, 9.0)
    p = lambda t: None if t is None else ptr(t)
    assert L.mi_mla_pre_gemm_i8(ptr(a8), M, 1536, ptr(wuq), Hq * 192, 1, None, p(bias), ptr(descale), p(rscale), ptr(y), code, stream_ptr()) == 0
    assert L.mi_mla_pre_bmm_rope(ptr(y), M, Hq, ptr(wuk_t), ptr(cos), ptr(sin), code, ptr(q0a), ptr(q1a), None, stream_ptr()) == 0
    assert L.mi_mla_pre_gemm2_bmm_rope(ptr(a8), M, ptr(wuq), Hq, p(bias), ptr(descale), p(rscale), ptr(wuk_t), ptr(cos), ptr(sin), code,
                                       ptr(q0b), ptr(q1b), None, stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(q1a.view(torch.int16), q1b.view(torch.int16)), "rope columns"
    assert torch.equal(q0a.view(torch.int16), q0b.view(torch.int16)), "BMM output"
