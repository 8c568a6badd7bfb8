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


@pytest.mark.parametrize("N,Hq,hidden", [(1, 32, 7168), (16, 64, 7168), (31, 128, 7168), (31, 128, 6144), (70, 16, 2048)])
def test_mla_preprocess(N, Hq, hidden):
    """torch.ops.npu.mla_preprocess vs the transcription of the reference golden (seed 42, shapes of
    tests/python/sgl_kernel_npu/test_mla_preprocess.py:487-498)."""
    torch.manual_seed(42)
    dt = torch.bfloat16
    block_size, nblocks = 128, 4
    hid = (torch.randn(N, hidden) * 0.5).to(dt)
    wdqkv = torch.randint(-8, 8, (2112, hidden), dtype=torch.int8)
    wuq = torch.randint(-8, 8, (Hq * 192, 1536), dtype=torch.int8)
    descale0 = (torch.rand(2112) * 1e-3 + 5e-4).float()
    descale1 = (torch.rand(Hq * 192) * 1e-3 + 5e-4).float()
    bias0 = torch.randint(-50, 50, (2112,), dtype=torch.int32)
    bias1 = torch.randint(-50, 50, (Hq * 192,), dtype=torch.int32)
    gamma0, beta0 = torch.randn(hidden).to(dt), torch.randn(hidden).to(dt)
    gamma1, beta1 = torch.randn(1536).to(dt), (torch.randn(1536) * 0.1).to(dt)
    gamma2 = torch.randn(512).to(dt)
    wuk = (torch.randn(Hq, 128, 512) * 0.1).to(dt)
    cos, sin = torch.rand(N, 64).to(dt), torch.rand(N, 64).to(dt)
    qs0, qo0 = torch.tensor([0.02]).to(dt), torch.tensor([3], dtype=torch.int8)
    qs1, qo1 = torch.tensor([0.03]).to(dt), torch.tensor([-2], dtype=torch.int8)
    slots = torch.randperm(nblocks * block_size)[:N].to(torch.int32)
    want = OK.mla_preprocess(hid, wdqkv, descale0, bias0, gamma1, beta1, gamma2, wuq, descale1, bias1, wuk, cos, sin, qs0, qo0, qs1, qo1)
    d = lambda t: t.cuda()
    kv = torch.zeros((nblocks, block_size, 1, 512), dtype=dt, device="cuda")
    kr = torch.zeros((nblocks, block_size, 1, 64), dtype=dt, device="cuda")
    q0 = torch.empty((N, Hq, 512), dtype=dt, device="cuda")
    q1 = torch.empty((N, Hq, 64), dtype=dt, device="cuda")
    out = torch.ops.npu.mla_preprocess(d(hid), d(gamma0), d(beta0), d(wdqkv), d(descale0), d(gamma1), d(beta1), d(wuq), d(descale1),
                                       d(gamma2), d(cos), d(sin), d(wuk), kv, kr, d(slots), d(qs0), d(qo0), d(bias0), d(qs1), d(qo1),
                                       d(bias1), cache_mode="krope_ctkv", quant_mode="per_tensor_quant_asymm", q_out0=q0,
                                       kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
    assert out[0].data_ptr() == q0.data_ptr() and out[1].data_ptr() == kv.data_ptr()
    k_nope = kv.view(-1, 512)[slots.long().cuda()].cpu()
    k_pe = kr.view(-1, 64)[slots.long().cuda()].cpu()
    tol = dict(rtol=2 ** -6, atol=2e-2)      # bf16 outputs: one ulp on either side of the golden
    assert torch.allclose(k_nope.float(), want[2].float(), **tol)
    assert torch.allclose(k_pe.float(), want[3].float(), **tol)
    assert torch.allclose(q1.cpu().float(), want[1].float(), **tol)
    assert torch.allclose(q0.cpu().float(), want[0].float(), rtol=2 ** -5, atol=5e-2)      # through a K=128 bf16 BMM
    # untouched cache rows stay zero
    mask = torch.ones(nblocks * block_size, dtype=torch.bool)
    mask[slots.long()] = False
    assert not kv.view(-1, 512).cpu()[mask].any()


@pytest.mark.parametrize("T,Hq,Hk,D,R", [(32, 8, 1, 64, 32), (32, 8, 1, 32, 32), (17, 16, 1, 128, 64), (64, 32, 1, 128, 64),
                                         (5, 128, 1, 192, 64)])       # reference cases (test_fused_rope_qk_mqa.py:64-71) + MLA-sized
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
