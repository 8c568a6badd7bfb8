"""Supplementary single-GPU measurements (BASELINE configs C3 / C5 shapes per rank, elementwise primitives).
Writes one JSON object to stdout; W = 1 so all EP traffic is local (HBM), E = 32 experts = the per-rank expert count at EP=8."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist

def ev_time(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    # the same call queued back to back (no synchronisation in between): a lone event-timed launch carries ~5 us of launch latency between its
    # two events, a fifth of a 30-us streaming kernel; `queued_us` is what a layer stream sees
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): fn()
    b.record(); torch.cuda.synchronize()
    return {"p50_us": ts[len(ts) // 2], "p99_us": ts[min(len(ts) - 1, int(len(ts) * 0.99))], "min_us": ts[0], "queued_us": a.elapsed_time(b) * 1e3 / 50}

def main():
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    import deep_ep, sgl_kernel_npu
    from sgl_kernel_npu.activation.swiglu_quant import swiglu_quant
    from sgl_kernel_npu.norm.add_rmsnorm_bias import add_rmsnorm_bias
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkv_rmsnorm_rope
    out = {}
    H, K, E = 7168, 8, 32
    buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
    g = torch.Generator(device="cuda").manual_seed(0)
    # ---- C3 shapes per rank: low-latency dispatch / combine, 128 tokens
    T = 128
    x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
    idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
    w = torch.rand((T, K), generator=g, device="cuda")
    (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
    y = (rx.float() * rs[:, None]).to(torch.bfloat16)
    out["ll_dispatch_128tok"] = ev_time(lambda: buf.low_latency_dispatch(x, idx, T, E, use_fp8=True))
    # (behind a dispatch, as in a decode step: a combine that directly follows another combine takes the three-launch form, deep_ep.hpp)
    def pair_combine():
        (_, _), _, h2, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
        return buf.low_latency_combine(y, idx, w, h2)
    tp = ev_time(pair_combine)
    out["ll_dispatch_combine_pair_128tok"] = tp
    out["ll_combine_128tok"] = ev_time(lambda: buf.low_latency_combine(y, idx, w, handle))
    out["ll_combine_128tok"]["note"] = "a combine behind a combine: the three-launch form; the two-launch form is inside ll_dispatch_combine_pair_128tok"
    # ---- C5 shapes per rank: fused_deep_moe, DeepSeek-V3 (H=7168, 2I=4096), 32 local experts
    I = 2048
    w13 = torch.randint(-16, 16, (E, 2 * I, H), generator=g, device="cuda", dtype=torch.int8)
    w2 = torch.randint(-16, 16, (E, H, I), generator=g, device="cuda", dtype=torch.int8)
    s13 = torch.rand((E, 2 * I), generator=g, device="cuda") * 4e-4 + 1.5e-3
    s2 = torch.rand((E, H), generator=g, device="cuda") * 4e-4 + 1.5e-3
    for T in (128, 4096):
        x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
        idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
        w = torch.rand((T, K), generator=g, device="cuda")
        f = lambda: buf.fused_deep_moe(x, idx, w, w13, s13, w2, s2, T, E)
        f()
        buf.begin_profile(0, 10, "")
        for _ in range(10): f()
        buf.end_profile()
        prof = {k: ms / n * 1e3 for k, (n, ms) in buf.get_profile_summary().items()}
        r = ev_time(f, n=10, warm=2)
        ops = T * K * (H * 2 * I + I * H) * 2
        r["int8_TOPs"] = ops / (r["p50_us"] * 1e-6) / 1e12
        r["kernels_avg_us"] = prof
        r["gemm_TOPs"] = ops / ((prof.get("moe_gemm1_swiglu", 0) + prof.get("moe_gemm2", 0)) * 1e-6) / 1e12
        out[f"fused_deep_moe_{T}tok_32experts"] = r
    # ---- elementwise primitives (HBM-bound): GB/s on algorithmic bytes
    S, h = 32768, 4096
    xs = torch.randn((S, h), generator=g, device="cuda").to(torch.bfloat16)
    gl = torch.full((32,), S // 32, dtype=torch.int64, device="cuda")
    t = ev_time(lambda: swiglu_quant(xs, gl, 1))
    out["swiglu_quant_32768x4096"] = dict(t, GBps=S * (h * 2 + h // 2 + 4) / t["p50_us"] / 1e3)
    B = 4096
    a, r_ = torch.randn((B, H), generator=g, device="cuda").to(torch.bfloat16), torch.randn((B, H), generator=g, device="cuda").to(torch.bfloat16)
    wt, bs = torch.randn(H, device="cuda").to(torch.bfloat16), torch.randn(H, device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: add_rmsnorm_bias(a, r_, wt, bs, 1e-6))
    out["add_rmsnorm_bias_4096x7168"] = dict(t, GBps=B * H * 8 / t["p50_us"] / 1e3)
    qkv = torch.randn((B, 6144 + 2048), generator=g, device="cuda").to(torch.bfloat16)
    sn, cs = torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16), torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16)
    hw = torch.randn(128, device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: split_qkv_rmsnorm_rope(qkv, sn, cs, 6144, 1024, 128, 1e-6, hw, hw, hw, hw))
    out["split_qkv_rmsnorm_rope_4096x8192"] = dict(t, GBps=B * 8192 * 4 / t["p50_us"] / 1e3)
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkvgate_gemma_rmsnorm_rope
    xg = torch.randn((B, 2 * 4096 + 2 * 1024), generator=g, device="cuda").to(torch.bfloat16)
    sg, cg = torch.rand((B, 64), device="cuda").to(torch.bfloat16), torch.rand((B, 64), device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: split_qkvgate_gemma_rmsnorm_rope(xg, sg, cg, 4096, 1024, 128, 64, 1e-6, hw, hw))
    out["split_qkvgate_gemma_rmsnorm_rope_4096x10240"] = dict(t, GBps=B * 10240 * 4 / t["p50_us"] / 1e3)
    # ---- attention with sinks, GPT-OSS decode shape: 64 q / 8 kv heads of 64, 4096 keys, window 128 and no window
    from sgl_kernel_npu.attention.sinks_attention import attention_sinks_triton
    Bs, Hqs, Hkvs, Ds, pg, Ss = 128, 64, 8, 64, 128, 4096
    nbs = Bs * Ss // pg
    qs_ = torch.randn((Bs, Hqs * Ds), generator=g, device="cuda").to(torch.bfloat16)
    kcs, vcs = (torch.randn((nbs, pg, Hkvs, Ds), generator=g, device="cuda").to(torch.bfloat16) for _ in range(2))
    bts = torch.randperm(nbs, device="cuda").to(torch.int32).reshape(Bs, Ss // pg)
    lns = torch.full((Bs,), Ss, dtype=torch.int32, device="cuda")
    snk = torch.randn(Hqs, device="cuda")
    for wname, wsz in (("full", -1), ("window128", 128)):
        t = ev_time(lambda: attention_sinks_triton(qs_, kcs, vcs, snk, bts, lns, Ds ** -0.5, wsz, Hqs, Hkvs), n=20, warm=3)
        keys = Ss if wsz < 0 else wsz
        out[f"attention_sinks_decode_b128_h64kv8_d64_s4096_{wname}"] = dict(t, GBps=Bs * keys * Hkvs * Ds * 2 * 2 / t["p50_us"] / 1e3)
    # ---- row statistics / scalings at the reference tests' shapes (fp32) and a bf16 model shape
    from sgl_kernel_npu.norm.rmsnorm_split import fused_rsqrt_mul, fused_variance
    from sgl_kernel_npu.norm.rmsnorm_without_weight import fused_rmsnorm_without_weight
    xr = torch.randn((1, 8190, 2560), generator=g, device="cuda")
    wr, vr = torch.randn(2560, device="cuda"), torch.rand(8190, device="cuda") + 0.1
    t = ev_time(lambda: fused_rsqrt_mul(xr, vr, wr, 1e-6))
    out["fused_rsqrt_mul_8190x2560_f32"] = dict(t, GBps=8190 * 2560 * 8 / t["p50_us"] / 1e3)
    t = ev_time(lambda: fused_variance(xr))
    out["fused_variance_8190x2560_f32"] = dict(t, GBps=8190 * 2560 * 4 / t["p50_us"] / 1e3)
    xb = torch.randn((1, 16384, 7168), generator=g, device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: fused_rmsnorm_without_weight(xb, 1e-6))
    out["rmsnorm_without_weight_16384x7168_bf16"] = dict(t, GBps=16384 * 7168 * 4 / t["p50_us"] / 1e3)
    from sgl_kernel_npu.activation.swiglu_oai_quant import swiglu_oai_quant
    xo = torch.randn((16384, 5760), generator=g, device="cuda").to(torch.bfloat16)      # GPT-OSS expert intermediate 2880
    t = ev_time(lambda: swiglu_oai_quant(xo, 1.702, 7.0))
    out["swiglu_oai_quant_16384x5760_bf16"] = dict(t, GBps=16384 * (5760 * 2 + 2880 + 4) / t["p50_us"] / 1e3)
    # ---- sparse + causal prefill on per-query block tables (MiniMax-M3 shape: 16 q heads on 1 kv head, d = 128, top-16 + own block of 128 keys)
    from sgl_kernel_npu.attention.fia_blockq_attention import flash_prefill_bnsd_blockq_sparse_fia
    Tq, bsz, tk1 = 4096, 128, 17
    n_ctx = 8192
    pages_f = n_ctx // bsz + 1
    rtt = (torch.arange(n_ctx, device="cuda", dtype=torch.int32) + bsz)[None].contiguous()        # request 0: pages 1 ..
    kf = torch.randn((pages_f, bsz, 1, 128), generator=g, device="cuda").to(torch.bfloat16)
    vf = torch.randn((pages_f, bsz, 1, 128), generator=g, device="cuda").to(torch.bfloat16)
    qf = torch.randn((Tq, 16, 128), generator=g, device="cuda").to(torch.bfloat16)
    sl = torch.arange(n_ctx - Tq + 1, n_ctx + 1, device="cuda", dtype=torch.int32)                 # the last 4096 positions of the context
    own_b = (sl - 1) // bsz
    tki = torch.stack([torch.randperm(int(n_ctx // bsz) - 2, device="cuda")[:tk1 - 1].to(torch.int32) for _ in range(64)]).repeat(Tq // 64, 1)
    tki = torch.minimum(tki, (own_b - 1).clamp(min=0)[:, None])
    tki = torch.cat([tki, own_b[:, None]], dim=1).contiguous()
    reqs = torch.zeros(Tq, dtype=torch.int32, device="cuda")
    t = ev_time(lambda: flash_prefill_bnsd_blockq_sparse_fia(qf, kf, vf, tki[None], sl, reqs, rtt, bsz, None, pages_f, tk1), n=20, warm=3)
    out["fia_blockq_sparse_prefill_4096q_h16_d128_top16x128"] = dict(t, keys_per_query=(tk1 - 1) * bsz + bsz // 2,
                                                                       note="duplicate block ids allowed in this synthetic selection; KV read per query from L2 / MALL")
    from sgl_kernel_npu.moe.mul_add import mul_add
    from sgl_kernel_npu.kimi_k3.attn_residual import mix_fused
    ra, rb = torch.randn((16384, 7168), generator=g, device="cuda").to(torch.bfloat16), torch.randn((16384, 7168), generator=g, device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: mul_add(ra, rb, 2.5))
    out["mul_add_16384x7168_bf16"] = dict(t, GBps=16384 * 7168 * 6 / t["p50_us"] / 1e3)
    pf, bk = torch.randn((4096, 7168), generator=g, device="cuda").to(torch.bfloat16), torch.randn((4096, 8, 7168), generator=g, device="cuda").to(torch.bfloat16)
    cwv = torch.randn(7168, generator=g, device="cuda") * 0.05
    t = ev_time(lambda: mix_fused(pf, bk, 8, cwv, 1e-6))
    out["attn_residual_mix_4096tok_8blocks_h7168_bf16"] = dict(t, GBps=4096 * 7168 * 2 * (9 + 1) / t["p50_us"] / 1e3,
                                                              note="algorithmic bytes: nine rows read once + the output; the rows are read a second time out of L2")
    from sgl_kernel_npu.activation.situ import situ_and_mul, situ_and_mul_quant
    xs = torch.randn((16384, 12288), generator=g, device="cuda").to(torch.bfloat16)      # d = 6144: the largest the quantising form takes
    t = ev_time(lambda: situ_and_mul_quant(xs))
    out["situ_and_mul_quant_16384x12288_bf16"] = dict(t, GBps=16384 * (12288 * 2 + 6144 + 4) / t["p50_us"] / 1e3)
    t = ev_time(lambda: situ_and_mul(xs))
    out["situ_and_mul_16384x12288_bf16"] = dict(t, GBps=16384 * (12288 * 2 + 6144 * 2) / t["p50_us"] / 1e3)
    from sgl_kernel_npu.norm.split_qkv_tp_rmsnorm_rope import split_qkv_tp_rmsnorm_rope
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_mrope import triton_split_qkv_rmsnorm_mrope
    from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope_pos_cache_half_npu import split_qkv_rmsnorm_rope_pos_cache_half_npu
    xt = torch.randn((B, 6144 + 2048), generator=g, device="cuda").to(torch.bfloat16)
    wq_, wk_ = torch.randn(6144, device="cuda").to(torch.bfloat16), torch.randn(1024, device="cuda").to(torch.bfloat16)
    s2, c2 = torch.rand((B, 128), device="cuda").to(torch.bfloat16), torch.rand((B, 128), device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: split_qkv_tp_rmsnorm_rope(xt, c2, s2, 6144, 1024, 128, 1e-6, wq_, wk_, 128, 1, None))
    out["split_qkv_tp_rmsnorm_rope_4096x8192"] = dict(t, GBps=B * 8192 * 4 / t["p50_us"] / 1e3, launches=2)
    cs3 = torch.randn((3, B, 128), device="cuda").to(torch.bfloat16)
    t = ev_time(lambda: triton_split_qkv_rmsnorm_mrope(xt, hw, hw, cs3, 48, 8, 128, 1e-6, [24, 20, 20], True))
    out["split_qkv_rmsnorm_mrope_4096x8192"] = dict(t, GBps=B * 8192 * 4 / t["p50_us"] / 1e3)
    cache = torch.randn((8192, 128), device="cuda")
    posb = torch.randint(0, 8192, (B,), device="cuda")
    t = ev_time(lambda: split_qkv_rmsnorm_rope_pos_cache_half_npu(xt, posb, cache, 6144, 1024, 128, eps=1e-6, q_weight=hw, k_weight=hw))
    out["split_qkv_rmsnorm_rope_pos_cache_4096x8192"] = dict(t, GBps=B * 8192 * 4 / t["p50_us"] / 1e3)
    # ---- fused_rope_qk_mqa (A13): MLA-sized q [T, 128, 192] with the first 64 dims rotated + one shared key head; and a pure-rope shape
    from sgl_kernel_npu.norm.fused_rope_qk_mqa import fused_rope_qk_mqa
    for (Tq, Dq, Rq) in ((4096, 192, 64), (4096, 64, 64)):
        qq = torch.randn((Tq, 128, Dq), device="cuda").to(torch.bfloat16)
        kk = torch.randn((Tq, 1, Dq), device="cuda").to(torch.bfloat16)
        csq = torch.rand((Tq, Rq), device="cuda").to(torch.bfloat16)
        t = ev_time(lambda: fused_rope_qk_mqa(qq, kk, csq, Rq, True))
        out[f"fused_rope_qk_mqa_{Tq}x129x{Dq}_r{Rq}_neox"] = dict(t, GBps=Tq * 129 * Dq * 4 / t["p50_us"] / 1e3)
        del qq, kk, csq
    # ---- paged GQA decode (HBM-bound): Llama-70B-like (64 q / 8 kv heads, D=128) and the reference's 288/256 config
    from sgl_kernel_npu.attention.decode_attention import decode_gqa
    for name, Bq, Hq, Hkv, D, Dv, Sq in (("gqa_decode_b64_h64kv8_d128_s4096", 64, 64, 8, 128, 128, 4096),
                                          ("gqa_decode_b256_h64kv8_d128_s4096", 256, 64, 8, 128, 128, 4096),
                                          ("gqa_decode_b128_h128kv1_d288_s4096", 128, 128, 1, 288, 256, 4096)):
        page = 64
        nb = Bq * Sq // page
        q = torch.randn((Bq, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
        kc = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
        vc = kc[..., :Dv] if D != Dv else torch.randn((nb, page, Hkv, Dv), generator=g, device="cuda").to(torch.bfloat16)
        bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(Bq, Sq // page)
        lens = torch.full((Bq,), Sq, dtype=torch.int32, device="cuda")
        o = torch.empty((Bq, Hq, Dv), device="cuda", dtype=torch.bfloat16)
        # MFMA-heavy decode kernels: 200 warm-up calls let the clocks settle (bench.py times MLA decode the same way); `queued_us` = 100 calls
        # queued back to back (what the layers of a decode step see), p50 / p99 / min = single calls with a synchronisation after each
        call = lambda: decode_gqa(q, kc, vc, o, lens, D ** -0.5, page, bt)
        t = ev_time(call, n=30, warm=200)
        a, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): call()
        b2.record(); torch.cuda.synchronize()
        kv_bytes = Bq * Sq * Hkv * (D if D != Dv else D + Dv) * 2
        out[name] = dict(t, GBps=kv_bytes / t["p50_us"] / 1e3, queued_us=a.elapsed_time(b2) * 10.0, v_is_view_of_k=bool(D != Dv))
    # ---- mla_preprocess (decode: 128 tokens, DeepSeek-V3 shapes: hidden 7168, 128 heads)
    N, Hh = 128, 128
    dt = torch.bfloat16
    dd = dict(device="cuda")
    hid = (torch.randn(N, H, **dd) * 0.5).to(dt)
    wdqkv = torch.randint(-8, 8, (2112, H), dtype=torch.int8, **dd)
    wuq = torch.randint(-8, 8, (Hh * 192, 1536), dtype=torch.int8, **dd)
    descale0, descale1 = torch.rand(2112, **dd) * 1e-3 + 5e-4, torch.rand(Hh * 192, **dd) * 1e-3 + 5e-4
    bias0, bias1 = torch.randint(-50, 50, (2112,), dtype=torch.int32, **dd), torch.randint(-50, 50, (Hh * 192,), dtype=torch.int32, **dd)
    gamma0, beta0 = torch.randn(H, **dd).to(dt), torch.randn(H, **dd).to(dt)
    gamma1, beta1, gamma2 = torch.randn(1536, **dd).to(dt), torch.randn(1536, **dd).to(dt), torch.randn(512, **dd).to(dt)
    wuk = (torch.randn(Hh, 128, 512, **dd) * 0.1).to(dt)
    cos, sin = torch.rand(N, 64, **dd).to(dt), torch.rand(N, 64, **dd).to(dt)
    qs0, qo0 = torch.tensor([0.02], **dd).to(dt), torch.tensor([3], dtype=torch.int8, **dd)
    qs1, qo1 = torch.tensor([0.03], **dd).to(dt), torch.tensor([-2], dtype=torch.int8, **dd)
    slots = torch.randperm(4096, **dd)[:N].to(torch.int32)
    kv, kr = torch.zeros((32, 128, 1, 512), dtype=dt, **dd), torch.zeros((32, 128, 1, 64), dtype=dt, **dd)
    q0, q1 = torch.empty((N, Hh, 512), dtype=dt, **dd), torch.empty((N, Hh, 64), dtype=dt, **dd)
    f = lambda: torch.ops.npu.mla_preprocess(hid, gamma0, beta0, wdqkv, descale0, gamma1, beta1, wuq, descale1, gamma2, cos, sin, wuk, kv, kr,
                                             slots, qs0, qo0, bias0, qs1, qo1, bias1, cache_mode="krope_ctkv",
                                             quant_mode="per_tensor_quant_asymm", q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
    t = ev_time(f, n=30, warm=10)
    wbytes = 2112 * H + Hh * 192 * 1536 + Hh * 128 * 512 * 2
    out["mla_preprocess_128tok"] = dict(t, weight_GBps=wbytes / t["p50_us"] / 1e3,
                                        note="same weights every call: 70 MB stay in the 256 MB memory-side cache between calls")
    # the same op over SIX weight sets in rotation (420 MB > the memory-side cache): every call streams its weights from HBM, as one layer
    # of a 61-layer decode step does
    sets = [(wdqkv, wuq, wuk)] + [(torch.randint(-8, 8, (2112, H), dtype=torch.int8, **dd), torch.randint(-8, 8, (Hh * 192, 1536), dtype=torch.int8, **dd),
                                   (torch.randn(Hh, 128, 512, **dd) * 0.1).to(dt)) for _ in range(5)]
    turn = [0]
    def f_cold():
        w0, w1, w2 = sets[turn[0] % len(sets)]
        turn[0] += 1
        torch.ops.npu.mla_preprocess(hid, gamma0, beta0, w0, descale0, gamma1, beta1, w1, descale1, gamma2, cos, sin, w2, kv, kr,
                                     slots, qs0, qo0, bias0, qs1, qo1, bias1, cache_mode="krope_ctkv",
                                     quant_mode="per_tensor_quant_asymm", q_out0=q0, kv_cache_out0=kv, q_out1=q1, kv_cache_out1=kr)
    t = ev_time(f_cold, n=36, warm=12)
    out["mla_preprocess_128tok_cold_weights"] = dict(t, weight_GBps=wbytes / t["p50_us"] / 1e3)
    for v in out.values():      # bandwidth at the queued rate beside the single-call one
        if isinstance(v, dict) and v.get("GBps") and v.get("queued_us") and v.get("p50_us"):
            v["GBps_queued"] = v["GBps"] * v["p50_us"] / v["queued_us"]
    print(json.dumps(out))

main()
