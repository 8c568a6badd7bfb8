"""Measurement helper used by the repo-level bench.py (not part of the reference API; imports nothing from oracle/)."""
import time

import torch

HBM_PEAK_GBPS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0


def _mla_inputs(B=128, Hq=128, S=4096, page=64, seed=1234, ragged=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    maxp = (S + page - 1) // page
    nb = B * maxp
    q = torch.randn((B, Hq, 576), generator=g, device="cuda").to(torch.bfloat16)
    kn = torch.randn((nb, page, 1, 512), generator=g, device="cuda").to(torch.bfloat16)
    kr = torch.randn((nb, page, 1, 64), generator=g, device="cuda").to(torch.bfloat16)
    bt = torch.randperm(nb, generator=g, device="cuda").to(torch.int32).reshape(B, maxp)   # permuted physical pages
    lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
    if ragged:
        lens = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32)
    return q, kn, kr, bt, lens


def bench_mla_decode(steps=30, warmup=5):
    """BASELINE config C4: bs=128, 128 q-heads over one latent KV head, D=576 (512+64), page 64, seqlen 4096, bf16."""
    from sgl_kernel_npu.attention.decode_attention import decode_mla

    B, Hq, S, page = 128, 128, 4096, 64
    q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    sm = 576 ** -0.5
    for _ in range(warmup):
        decode_mla(q, kn, kr, out, lens, sm, page, bt)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    # The timed calls are the attention layers of decode steps of 61 layers (DeepSeek-V3): the layers of a step pass the same kv_seq_lens
    # tensor and share one work list (built by the step's first layer, csrc/pytorch_extensions.cpp: cached_mla_plan); a new step -- an
    # in-place write to kv_seq_lens, here at the first timed call and every 61 calls -- rebuilds it INSIDE the timed region.
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(evs):
        a.record()
        if i % 61 == 0:
            lens.add_(0)
        decode_mla(q, kn, kr, out, lens, sm, page, bt)
        b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    dev_ms = sum(a.elapsed_time(b) for a, b in evs) / steps
    # the ragged copy of C4 (kv_seq_lens ~ U[1, 4096], same pages): about half the keys -- reported beside the headline, not part of it.
    # Default = the planned form (a device-built, length-aware work list: split counts per sequence, longest pieces first); the same
    # batch with the uniform two splits of round 3 (the longest sequence sets the pace of its workgroups) is timed beside it.
    _, _, _, _, rlens = _mla_inputs(B, Hq, S, page, ragged=True)

    def timed(ls, num_splits=0):          # 0 = the library's choice (the Python entry point's), -1 = a list per call, n = uniform splits
        call = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, ls, float(sm), int(page), bt, num_splits)
        for _ in range(max(warmup // 4, 5)):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps):
            if i % 61 == 0:
                ls.add_(0)                # a new decode step: the shared list is rebuilt inside the timed region
            call()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    def timed_shared_plan(ls):            # plan once, run many: what the layers of one decode step pay when the step builds the list once
        from sgl_kernel_npu.attention.decode_attention import decode_mla_plan
        plan = decode_mla_plan(ls, 1)
        call = lambda: decode_mla(q, kn, kr, out, ls, sm, page, bt, plan=plan)
        for _ in range(max(warmup // 4, 5)):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            call()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    r_ms = timed(rlens)
    ms_per_call_plan = timed(lens, num_splits=-1)           # every call builds its own list (no sharing between layers)
    r_ms_per_call_plan = timed(rlens, num_splits=-1)
    r_ms_shared = timed_shared_plan(rlens)
    ms_shared = timed_shared_plan(lens)
    r_ms_uniform = timed(rlens, num_splits=2)
    ms_uniform = timed(lens, num_splits=2)
    # the same cache read by a 16-head shard (TP 8 of the 128 heads: what one rank of a tensor-parallel deployment runs), the 64-head
    # kernel: planned (default) against two uniform splits
    q16 = q[:, :16].contiguous()
    out16 = torch.empty((B, 16, 512), dtype=torch.bfloat16, device="cuda")

    def timed16(ls, num_splits=0):
        call = lambda: torch.ops.npu.decode_mla(q16, kn, kr, out16, ls, float(sm), int(page), bt, num_splits)
        for _ in range(max(warmup // 4, 5)):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(steps):
            if i % 61 == 0:
                ls.add_(0)
            call()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    shard = {"workload": "the C4 cache read by a 16-head shard (TP 8), 64-head kernel", "ms_per_step": timed16(lens), "ragged_ms_per_step": timed16(rlens),
             "uniform_2_splits_ms_per_step": timed16(lens, 2), "ragged_uniform_2_splits_ms_per_step": timed16(rlens, 2)}
    shard["frac"] = (float(lens.sum().item()) * 1152 + B * 16 * 2176) / (shard["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
    shard["ragged_frac"] = (float(rlens.sum().item()) * 1152 + B * 16 * 2176) / (shard["ragged_ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
    r_bytes = float(rlens.sum().item()) * 576 * 2 + B * Hq * (576 + 512) * 2
    kv_bytes = float(lens.sum().item()) * 576 * 2
    io_bytes = B * Hq * (576 + 512) * 2
    flops = float(lens.sum().item()) * Hq * (576 + 512) * 2
    achieved = (kv_bytes + io_bytes) / (dev_ms * 1e-3) / 1e9
    return {
        "metric": "MLA decode tok/s", "value": B / (dev_ms * 1e-3), "unit": "tok/s", "ms_per_step": dev_ms,
        "host_ms_per_step": wall * 1e3, "dtype": "bf16",
        "config": {"workload": "MLA paged decode, bs=128, q_heads=128, kv_heads=1, head_dim=576 (512+64), page_size=64, "
                               "seqlen=4096 (BASELINE C4)"},
        "roofline": {"bound": "hbm", "kernel": "decode_plan_kernel (once per 61-call step) + mla_decode_wide8s_kernel (two-piece sequences finish in the kernel) + mla_merge_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes": kv_bytes + io_bytes, "avg_launch_us": dev_ms * 1e3},
        "ragged": {"workload": "same batch, kv_seq_lens ~ U[1, 4096]", "ms_per_step": r_ms, "mean_seq_len": float(rlens.float().mean().item()),
                   "achieved_GBps": r_bytes / (r_ms * 1e-3) / 1e9, "frac": r_bytes / (r_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "per_call_plan_ms_per_step": r_ms_per_call_plan,
                   "shared_plan_ms_per_step": r_ms_shared, "shared_plan_frac": r_bytes / (r_ms_shared * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                   "uniform_2_splits_ms_per_step": r_ms_uniform, "uniform_2_splits_frac": r_bytes / (r_ms_uniform * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        # the same batch with the work list built ONCE outside the loop (decode_mla_plan; the layers of a decode step share it) and passed in
        "per_call_plan_ms_per_step": ms_per_call_plan,      # num_splits = -1: no sharing between the layers of a step
        "plan_sharing": "the layers of a decode step (61 calls on one kv_seq_lens tensor) share one work list, rebuilt inside the timed region at every step",
        "shared_plan_ms_per_step": ms_shared, "shared_plan_frac": (kv_bytes + io_bytes) / (ms_shared * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "tp8_shard": shard,
        "uniform_2_splits_ms_per_step": ms_uniform,     # the full-length batch through num_splits = 2 (round 3's form), queued back to back
        "pmc_kernels": ["decode_plan_kernel", "mla_decode_wide8s_kernel<true, true>", "mla_merge_kernel<true>"],     # launches of one step (bench.py looks up their PMC traffic)
        "mfma": {"achieved_TFLOPs": flops / (dev_ms * 1e-3) / 1e12, "peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
                 "frac": flops / (dev_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
    }
