#!/usr/bin/env python3
"""Hot-path benchmark: DeepEP normal dispatch (INT8) + combine (BF16) on MI355X, BASELINE.json config C2 shapes.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One step = one pass of the hot path over one batch of synthetic tokens on every rank:
    get_dispatch_layout -> dispatch(quant_mode="int8") -> combine(BF16)          (deep_ep.Buffer API, EP = N ranks)
with T = 4096 tokens/rank, hidden 7168, top-8 of 256 experts (DeepSeek-V3 shapes); inputs are resident in HBM.
Weak scaling: per-GPU work is fixed as N grows.

`value` follows the reference's own bandwidth convention (tests/python/deepep/test_intranode.py:447-448,530-534):
bytes = BF16-equivalent size of every received row (local rows included) for dispatch plus the same for combine,
summed over all ranks, divided by the max-over-ranks step time.  At N = 1 everything is a local permutation and the
kernels are HBM-bound; the `roofline` object prices the dominant kernel against HBM.  A second object `mla_decode`
reports BASELINE config C4 (single-GPU MLA paged decode) when that kernel is built.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "sgl-kernel-npu_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
XGMI_LINK_GBPS = 153.0          # per link, 7 links per GPU
T_TOKENS, HIDDEN, TOPK, EXPERTS = 4096, 7168, 8, 256


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=T_TOKENS)
    ap.add_argument("--strategy", default=os.getenv("DEEP_BENCH_STRATEGY", "default"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mla", action="store_true")
    return ap.parse_args()


def init_dist(n):
    if n > 1 or "RANK" in os.environ:
        rank = int(os.environ.get("RANK", 0))
        world = int(os.environ.get("WORLD_SIZE", n))
        local = int(os.environ.get("LOCAL_RANK", rank))
        backend = "nccl"
        if os.environ.get("BENCH_SINGLE_DEVICE") == "1":      # test hook: every rank on cuda:0 over gloo (the launch / JSON / fallback
            local, backend = 0, "gloo"                         # plumbing of the N > 1 path on a one-GPU box; not a measurement)
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 500))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist.get_rank(), dist.get_world_size()


def make_inputs(rank, T):
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)      # seed = 1234 + rank
    x = torch.randn((T, HIDDEN), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    scores = torch.randn((T, EXPERTS), generator=g, device="cuda").abs() + 1
    topk_idx = torch.topk(scores, TOPK, dim=-1, largest=True, sorted=False)[1]
    topk_w = torch.randn((T, TOPK), generator=g, device="cuda", dtype=torch.float32)
    return x, topk_idx, topk_w


def one_step(buf, x, topk_idx, topk_w, y):
    per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(topk_idx, EXPERTS)
    recv, _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                              num_tokens_per_expert=per_expert, topk_idx=topk_idx, topk_weights=topk_w,
                                              quant_mode="int8")
    n = sum(lst)
    if y is None or y.shape[0] < max(n, 1):
        # expert stand-in: de-quantised rows (reference test convention per_token_cast_back), made once, untimed
        y = (recv[0].float() * recv[1][:, None]).to(torch.bfloat16)
    out, _, _ = buf.combine(y, handle)
    return out, n, y, recv, handle


def flush_c_stdout():
    """RCCL printf()s its version banner at communicator creation; with stdout a pipe that text waits in libc's buffer until
    exit -- i.e. after the JSON line -- unless it is pushed out first (every rank, right after the first collective)."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass


def barrier_sync():
    dist.barrier()
    torch.cuda.synchronize()


def check_round_trip(out, x, topk_w):
    golden = x.float() * topk_w.sum(dim=1, keepdim=True)
    a, b = out.double() + 1, golden.double() + 1
    return float(1 - 2 * (a * b).sum() / (a * a + b * b).sum())


def kernel_bytes(name, T, K, H, n_pairs, n_recv):
    """Algorithmic HBM bytes of one launch."""
    row = H + 16
    return {
        "dispatch_stage": T * H * 2 + T * row + n_pairs * 8,    # read bf16 tokens once, write one int8 row per token + the index
        "dispatch_pull": 2 * n_recv * row + n_recv * 8,         # read a token row + index entry per received row, write recv_x / scales / triples
        "combine_push": 2 * n_recv * H * 2,                     # read bf16 rows, write them into the owners' slots
        "combine_reduce": n_pairs * H * 2 + T * H * 2,          # read K slots per token, write one bf16 row
    }[name]


PMC_KERNEL_NAMES = {"dispatch_stage": "stage_int8_kernel<false, true>", "dispatch_pull": "pull_indexed_kernel",
                    "combine_push": "combine_push_kernel", "combine_reduce": "combine_reduce_kernel<false, 8>"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of this same command
    (profiles/r*_pmc_traffic.json, produced by tools/summarize_prof.py with the gfx950 FETCH_SIZE x2 correction);
    None when no PMC summary is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(PMC_KERNEL_NAMES.get(kernel, kernel))
        return k["hbm_bytes_per_launch"] if k else None
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(sample_tokens):
    """The oracle (a port of the reference arithmetic, NumPy, 1 thread of compute) on a bounded sample of the same
    workload: W = 1, `sample_tokens` tokens, same hidden / top-k / experts; dispatch(int8) + cast-back + combine."""
    import numpy as np

    from oracle import ep as O
    from oracle.bf16 import f32_to_bf16_bits_rne

    rng = np.random.default_rng(0)
    x = f32_to_bf16_bits_rne(rng.standard_normal((sample_tokens, HIDDEN)).astype(np.float32))
    scores = np.abs(rng.standard_normal((sample_tokens, EXPERTS))) + 1
    idx = np.argpartition(-scores, TOPK, axis=1)[:, :TOPK].astype(np.int64)
    w = rng.standard_normal((sample_tokens, TOPK)).astype(np.float32)
    t0 = time.perf_counter()
    reps = 0
    while True:                       # ~10 s of CPU work, whole passes only
        res = O.normal_dispatch([x], [idx], EXPERTS, quant=True)[0]
        y = O.per_token_cast_back(res.recv_x, res.recv_x_scales)
        O.combine([y], [res.recv_src_idx], [res.total_recv], [idx], [w], EXPERTS)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 64:
            break
    bytes_ = 2 * res.total_recv * HIDDEN * 2 * reps
    return {"value": bytes_ / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} passes of W=1, {sample_tokens} tokens x hidden {HIDDEN} x top-{TOPK} of {EXPERTS} experts: "
                      f"layout + int8 dispatch + cast-back + bf16 combine through oracle/ep.py (NumPy), {dt:.2f} s",
            "seconds": dt}


def mla_section(args):
    try:
        from sgl_kernel_npu.bench_hooks import bench_mla_decode     # present once the MLA kernel is built
    except Exception:
        return None
    try:
        r = bench_mla_decode(steps=100, warmup=300)      # MFMA-heavy: let the clocks settle (tens of ms)
        parts = [pmc_traffic(k) for k in ("mla_decode_wide_kernel<true>", "mla_merge_kernel<true>")]
        if all(v is not None for v in parts):
            r["roofline"]["traffic"] = sum(parts)      # both launches of one decode step (split partials included)
        return r
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main():
    args = parse()
    rank, world = init_dist(args.gpus)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    import deep_ep

    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "10000")     # bounded spins: a broken peer mapping fails fast, then falls back
    T = args.tokens
    x, topk_idx, topk_w = make_inputs(rank, T)
    group = dist.group.WORLD
    strategy = args.strategy
    buf = deep_ep.Buffer(group, normal_strategy=strategy, low_latency_strategy=strategy)
    validated = None
    try:
        out, n_recv, y, recv, handle = one_step(buf, x, topk_idx, topk_w, None)
        torch.cuda.synchronize()
        validated = check_round_trip(out, x, topk_w) < 3e-3          # reference threshold for int8 (utils.py:198-203)
    except Exception as e:  # noqa: BLE001
        if rank == 0:
            print(f"[bench] strategy {strategy} failed ({e}); falling back to alltoall", file=sys.stderr)
        validated = False
    flag = torch.tensor([1 if validated else 0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    flush_c_stdout()
    if int(flag.item()) == 0 and strategy != "alltoall":
        strategy = "alltoall"
        buf = deep_ep.Buffer(group, normal_strategy="alltoall", low_latency_strategy="alltoall")
        out, n_recv, y, recv, handle = one_step(buf, x, topk_idx, topk_w, None)
        torch.cuda.synchronize()
        validated = check_round_trip(out, x, topk_w) < 3e-3
    strategy = buf.normal_strategy.get_name()

    for _ in range(args.warmup):
        one_step(buf, x, topk_idx, topk_w, y)
    profiled = hasattr(buf.runtime, "get_profile_summary") and strategy == "default"
    if profiled:
        buf.begin_profile(0, args.steps, "")
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(buf, x, topk_idx, topk_w, y)
    barrier_sync()
    dt = time.perf_counter() - t0
    prof = {}
    if profiled:
        buf.end_profile()
        prof = buf.get_profile_summary()
    tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rows = torch.tensor([n_recv], device="cuda", dtype=torch.float64)
    dist.all_reduce(rows, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_rows = float(rows.item())
    ms_per_step = dt / args.steps * 1e3
    bytes_per_step = 2 * total_rows * HIDDEN * 2          # dispatch recv + combine send, BF16-equivalent (reference convention)
    value = bytes_per_step / (ms_per_step * 1e-3) / 1e9

    if rank != 0:
        dist.destroy_process_group()
        flush_c_stdout()
        return
    n_pairs = int((topk_idx >= 0).sum().item())
    result = {
        "metric": "dispatch+combine GB/s (EP=N, 4096 tok/rank, h=7168, top-8, INT8 dispatch / BF16 combine; "
                  "reference convention: BF16-equivalent received rows / time)",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"deep_ep normal dispatch(int8)+combine(bf16), EP={world}, {T} tok/rank, hidden {HIDDEN}, "
                               f"top-{TOPK} of {EXPERTS} experts (BASELINE C2 shapes at EP={world})",
                   "strategy": strategy, "tokens_per_rank": T, "hidden": HIDDEN, "topk": TOPK, "experts": EXPERTS},
        "per_gpu_GBps": value / world, "validated_round_trip": bool(validated),
    }
    if prof:
        per = {k: {"launches": n, "avg_us": ms / n * 1e3} for k, (n, ms) in prof.items() if n}
        dom = max(per, key=lambda k: per[k]["avg_us"])
        alg = kernel_bytes(dom, T, TOPK, HIDDEN, n_pairs, n_recv)
        achieved = alg / (per[dom]["avg_us"] * 1e-6) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(dom) if world == 1 else None,
                              "algorithmic_bytes": alg,
                              "avg_launch_us": per[dom]["avg_us"]}
        result["kernels"] = {k: dict(v, GBps=kernel_bytes(k, T, TOPK, HIDDEN, n_pairs, n_recv) / (v["avg_us"] * 1e-6) / 1e9)
                             for k, v in per.items() if k in ("dispatch_stage", "dispatch_pull", "combine_push", "combine_reduce")}
        if world > 1:
            # egress over xGMI per GPU: rows whose expert lives on another rank
            remote = n_recv * (world - 1) / world
            t_pull = per.get("dispatch_pull", {}).get("avg_us", 0) * 1e-6
            t_push = per.get("combine_push", {}).get("avg_us", 0) * 1e-6
            peak = XGMI_LINK_GBPS * (world - 1)
            result["xgmi"] = {
                "peak_GBps": peak,
                "dispatch_GBps": remote * (HIDDEN + 4) / t_pull / 1e9 if t_pull else None,
                "combine_GBps": remote * HIDDEN * 2 / t_push / 1e9 if t_push else None,
            }
            for k in ("dispatch_GBps", "combine_GBps"):
                if result["xgmi"][k]:
                    result["xgmi"][k.replace("GBps", "frac")] = result["xgmi"][k] / peak
    if world == 1 and not args.no_mla:          # before the CPU leg: the GPU clocks sag while the host works alone
        mla = mla_section(args)
        if mla is not None:
            result["mla_decode"] = mla
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(T)
    # The JSON line must be the last thing on stdout.  RCCL printf()s its version banner at communicator creation; with stdout
    # a pipe that text waits in libc's buffer until exit -- i.e. after anything Python prints -- unless it is flushed first.
    torch.cuda.synchronize()
    dist.destroy_process_group()
    sys.stderr.flush()
    flush_c_stdout()
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
