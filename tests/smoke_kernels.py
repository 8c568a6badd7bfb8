"""One tiny invocation of every sgl_kernel_npu primitive on cuda:0 against the CPU oracle (used by __graft_entry__.smoke();
test infrastructure: lives outside the product package because it imports oracle/)."""
import torch


def smoke_kernels():
    """One tiny invocation of every primitive on cuda:0 against the CPU oracle."""
    from oracle import kernels as OK
    from sgl_kernel_npu.attention.decode_attention import decode_mla

    torch.manual_seed(0)
    B, Hq, S, page = 2, 16, 100, 32
    maxp = (S + page - 1) // page
    q = torch.randn((B, Hq, 576)).to(torch.bfloat16)
    kn = torch.randn((B * maxp, page, 1, 512)).to(torch.bfloat16)
    kr = torch.randn((B * maxp, page, 1, 64)).to(torch.bfloat16)
    bt = torch.randperm(B * maxp).to(torch.int32).reshape(B, maxp)
    lens = torch.tensor([S, S - 30], dtype=torch.int32)
    want = OK.decode_mla(q, kn, kr, lens, bt, 576 ** -0.5)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    decode_mla(q.cuda(), kn.cuda(), kr.cuda(), out, lens.cuda(), 576 ** -0.5, page, bt.cuda())
    assert torch.allclose(out.cpu().float(), want.float(), atol=1e-3, rtol=2 ** -7), "MLA decode differs from the oracle"
    # 128 heads on one latent head: the wide (32x32x16 MFMA) kernel + merge
    q2 = torch.randn((B, 128, 576)).to(torch.bfloat16)
    want2 = OK.decode_mla(q2, kn, kr, lens, bt, 576 ** -0.5)
    out2 = torch.empty((B, 128, 512), dtype=torch.bfloat16, device="cuda")
    decode_mla(q2.cuda(), kn.cuda(), kr.cuda(), out2, lens.cuda(), 576 ** -0.5, page, bt.cuda())
    assert torch.allclose(out2.cpu().float(), want2.float(), atol=1e-2, rtol=1e-2), "wide MLA decode differs from the oracle"
    print("[smoke] decode_mla matches the oracle (64-head and 128-head kernels)")
    from sgl_kernel_npu.activation.swiglu_quant import swiglu_quant
    from sgl_kernel_npu.attention.decode_attention import decode_gqa
    from sgl_kernel_npu.norm.add_rmsnorm_bias import add_rmsnorm_bias
    from sgl_kernel_npu.norm.fused_rope_qk_mqa import fused_rope_qk_mqa

    kg = torch.randn((B * maxp, page, 2, 128)).to(torch.bfloat16)
    vg = torch.randn((B * maxp, page, 2, 128)).to(torch.bfloat16)
    qg = torch.randn((B, 16, 128)).to(torch.bfloat16)
    wg = OK.decode_gqa(qg, kg, vg, lens, bt, 128 ** -0.5)
    og = torch.empty((B, 16, 128), dtype=torch.bfloat16, device="cuda")
    decode_gqa(qg.cuda(), kg.cuda(), vg.cuda(), og, lens.cuda(), 128 ** -0.5, page, bt.cuda())
    assert torch.allclose(og.cpu().float(), wg.float(), atol=1e-2, rtol=1e-2), "GQA decode differs from the oracle"
    xs = (torch.randn((40, 512)) * 2).to(torch.bfloat16)
    gl = torch.tensor([10, 0, 25], dtype=torch.int64)
    wq, wsc, tot = OK.swiglu_quant(xs, gl, 1)
    gq, gsc = swiglu_quant(xs.cuda(), gl.cuda(), 1)
    assert (gq[:tot].cpu().int() - wq[:tot].int()).abs().max() <= 1 and torch.allclose(gsc[:tot].cpu(), wsc[:tot], rtol=5e-3)
    xa, ra, wa, ba = (torch.randn(6, 1024).to(torch.bfloat16) for _ in range(2)), None, torch.randn(1024).to(torch.bfloat16), None
    xa, ra = xa
    w1, w2 = OK.add_rmsnorm_bias(xa, ra, wa, None, 1e-6)
    o1, o2 = add_rmsnorm_bias(xa.cuda(), ra.cuda(), wa.cuda(), None, 1e-6)
    assert torch.equal(o2.cpu(), w2) and torch.allclose(o1.cpu().float(), w1.float(), rtol=2 ** -7, atol=1e-3)
    qr, kr2, cs = torch.randn(9, 8, 64).to(torch.float16), torch.randn(9, 1, 64).to(torch.float16), torch.randn(9, 32).to(torch.float16)
    wqr, wkr = OK.fused_rope_qk_mqa(qr, kr2, cs, 32, True)
    gqr, gkr = fused_rope_qk_mqa(qr.cuda(), kr2.cuda(), cs.cuda(), 32, True)
    assert torch.equal(gqr.cpu(), wqr) and torch.equal(gkr.cpu(), wkr)
    print("[smoke] decode_gqa, swiglu_quant, add_rmsnorm_bias, fused_rope_qk_mqa match the oracle")
