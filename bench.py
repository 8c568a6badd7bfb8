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
kernels are HBM-bound; the `roofline` object prices the dominant kernel against HBM.  At N > 1 both dispatch transports
(push = remote writes, pull = remote reads) are timed with the same K steps; the faster one is the headline, the other is
reported beside it under `transports`, and `xgmi` prices the cross-GPU legs against 153 GB/s per link.

Further objects on the same JSON line (the other BASELINE configs, measured by the same process):
  low_latency     C3: low-latency dispatch / combine at 128 tokens per rank, p50 / p99 per call (HIP events)
  fused_deep_moe  C5: dispatch -> INT8 grouped GEMM1 + SwiGLU -> requant -> GEMM2 -> combine, 4096 tokens per rank, 32 local experts
  mla_decode      C4: MLA paged decode (N = 1 only) with its own roofline and cpu_baseline
  cpu_baseline    N = 1 only: the reference's alltoall algorithm restated on the host (oracle/cpu_alltoall.py, 8 gloo ranks
                  sharing all host cores) on the same C2 workload, plus the single-core NumPy oracle.
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
INT8_PEAK_TOPS = 3900.0         # dense int8 MFMA peak used in DESIGN.md (half the fp8 figure is quoted for bf16)
T_TOKENS, HIDDEN, TOPK, EXPERTS = 4096, 7168, 8, 256
INTER = 2048                    # DeepSeek-V3 MoE intermediate size (2I = 4096)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=T_TOKENS)
    ap.add_argument("--strategy", default=os.getenv("DEEP_BENCH_STRATEGY", "default"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mla", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 (low-latency) and C5 (fused_deep_moe) sections")
    ap.add_argument("--dry-run-8", action="store_true",
                    help="plumbing check of the 8-GPU launch on ONE GPU: re-executes itself as 8 ranks that all use cuda:0 (gloo bootstrap, "
                         "hipIpc windows), full C2 size; the JSON line carries every N > 1 key and \"dry_run_single_device\": true -- NOT a measurement")
    return ap.parse_args()


def self_launch_command(args, environ):
    """The driver starts N > 1 through `python -m torch.distributed.run ... bench.py --gpus N ...` (RANK / WORLD_SIZE set), but a plain
    `python bench.py --gpus N` must produce the same line instead of waiting in a rendezvous nobody else joins: when --gpus N > 1 and
    the environment carries no RANK, this process re-executes itself through torch.distributed.run, one rank per GPU.  `--dry-run-8` is
    the same re-execution with every rank on cuda:0 (BENCH_SINGLE_DEVICE=1).  -> (argv, env) to exec, or None to run in this process."""
    if "RANK" in environ:
        return None                                   # already a rank of a launcher (the driver's, or our own re-execution)
    n = 8 if args.dry_run_8 else args.gpus
    if n <= 1:
        return None
    env = dict(environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if args.dry_run_8:
        env["BENCH_SINGLE_DEVICE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.abspath(__file__), "--gpus", str(n), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--tokens", str(args.tokens), "--strategy", args.strategy]
    for flag, on in (("--dry-run-8", args.dry_run_8), ("--no-extra", args.no_extra), ("--no-mla", args.no_mla),
                     ("--no-cpu-baseline", args.no_cpu_baseline)):
        if on:
            cmd.append(flag)
    return cmd, env


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
        import datetime
        tmo = datetime.timedelta(seconds=300)      # a desynchronised rank ends the run in minutes, not after the default half hour
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=tmo)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
    else:
        torch.cuda.set_device(0)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 500))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist.get_rank(), dist.get_world_size()


def make_inputs(rank, T, experts=EXPERTS):
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)      # seed = 1234 + rank
    x = torch.randn((T, HIDDEN), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    scores = torch.randn((T, experts), generator=g, device="cuda").abs() + 1
    topk_idx = torch.topk(scores, TOPK, dim=-1, largest=True, sorted=False)[1]
    topk_w = torch.randn((T, TOPK), generator=g, device="cuda", dtype=torch.float32)
    return x, topk_idx, topk_w


def one_step(buf, x, topk_idx, topk_w, y):
    per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(topk_idx, EXPERTS)
    recv, _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                              num_tokens_per_expert=per_expert, topk_idx=topk_idx, topk_weights=topk_w,
                                              quant_mode="int8")
    n = recv[0].shape[0]          # == sum(lst): rows received (kept O(1): the host is on the critical path between dispatch and combine)
    if y is None or y.shape[0] < max(n, 1):
        # expert stand-in: de-quantised rows (reference test convention per_token_cast_back), made once, untimed
        y = (recv[0].float() * recv[1][:, None]).to(torch.bfloat16)
    out, _, _ = buf.combine(y, handle)
    return out, n, y, recv, handle


def one_step_nosync(buf, x, topk_idx, topk_w, y, worst):
    """The same step in DeepEP's graph-friendly form: dispatch(num_worst_tokens = worst) returns worst-case sized outputs and the host
    never learns the row count (no pinned-word spin, no allocation after the exchange); combine consumes the handle as is."""
    per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(topk_idx, EXPERTS)
    recv, _, _, _, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in, num_tokens_per_expert=per_expert,
                                            topk_idx=topk_idx, topk_weights=topk_w, quant_mode="int8", num_worst_tokens=worst)
    if y is None:
        y = (recv[0].float() * recv[1][:, None]).to(torch.bfloat16)      # expert stand-in, made once, untimed
    out, _, _ = buf.combine(y, handle)
    return out, y


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


_FLUSH = None


def flush_cache():
    """The reference's timer writes a 256 MB buffer before the timed region so nothing is served from a warm cache
    (tests/python/deepep/utils.py:58-93); 256 MB also covers the MI355X's 256 MB MALL."""
    global _FLUSH
    if _FLUSH is None:
        _FLUSH = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    _FLUSH.zero_()


def check_round_trip(out, x, topk_w):
    golden = x.float() * topk_w.sum(dim=1, keepdim=True)
    a, b = out.double() + 1, golden.double() + 1
    return float(1 - 2 * (a * b).sum() / (a * a + b * b).sum())


def ev_stats(fn, n=50, warm=10, pre=None):
    """Per-call device time of fn() from HIP events on the current stream -> dict(p50_us, p99_us, min_us).  pre(): an UNTIMED call queued in
    front of every timed one (the start event is recorded behind it)."""
    for _ in range(warm):
        if pre:
            pre()
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if pre:
            pre()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return {"p50_us": ts[len(ts) // 2], "p99_us": ts[min(len(ts) - 1, int(len(ts) * 0.99))], "min_us": ts[0]}


def queued_stats(fn, n=200, warm=10, pre=None):
    """As ev_stats, but the n calls are queued without a host synchronisation in between (one at the end): the GPU never idles between
    calls, as in a serving loop that streams layers, so its clocks stay up and launch latency hides behind the previous call."""
    for _ in range(warm):
        if pre:
            pre()
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    host = []
    for a, b in evs:
        if pre:
            pre()
        a.record()
        h0 = time.perf_counter()
        fn()
        host.append((time.perf_counter() - h0) * 1e6)
        b.record()
    torch.cuda.synchronize()
    dev = [a.elapsed_time(b) * 1e3 for a, b in evs]
    ts = sorted(dev)
    # The slowest device samples are not slow kernels: the event pair brackets "start event executed ... end event executed" on the GPU,
    # so whenever the HOST takes longer to enqueue a call than the GPU takes to run it (allocator growth, a Python GC pause) the GPU sits
    # between the two events waiting for work.  Reported so that the tail can be read: the host's own enqueue time of the slowest sample.
    worst = max(range(n), key=lambda i: dev[i])
    hs = sorted(host)
    return {"p50_us": ts[len(ts) // 2], "p99_us": ts[min(len(ts) - 1, int(len(ts) * 0.99))], "min_us": ts[0],
            "host_enqueue_us_p50": hs[len(hs) // 2], "host_enqueue_us_max": hs[-1], "host_enqueue_us_of_slowest_sample": host[worst],
            "slowest_sample_index": worst}


def graph_stats(fn, n=200, warm=3):
    """fn() captured ONCE in a HIP graph (torch.cuda.graph), then per-replay device time from HIP events: what a serving stack that
    captures its decode step sees -- no host work between the kernels of a call.  Every rank must call this at the same point (the
    captured calls are collective).  -> dict(p50_us, p99_us, min_us)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()                       # outputs live in the graph's pool  # noqa: F841
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return {"p50_us": ts[len(ts) // 2], "p99_us": ts[min(len(ts) - 1, int(len(ts) * 0.99))], "min_us": ts[0]}


def max_over_ranks(d):
    """Element-wise max over ranks of a flat {name: float} dict (every rank calls it with the same keys)."""
    keys = sorted(d)
    t = torch.tensor([float(d[k]) for k in keys], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return {k: float(v) for k, v in zip(keys, t.tolist())}


def routing_stats(topk_idx, world, rank):
    """Per-destination counts of this rank's routing (all device-side, one sync): pairs[d] = (t, k) selections owned by rank d,
    tokens[d] = distinct tokens with at least one expert on rank d."""
    L = EXPERTS // world
    dest = topk_idx // L
    valid = topk_idx >= 0
    pairs = torch.bincount(dest[valid].reshape(-1), minlength=world)
    onehot = torch.zeros((topk_idx.shape[0], world), dtype=torch.bool, device=topk_idx.device)
    onehot.scatter_(1, dest.clamp(min=0), valid)
    tokens = onehot.sum(dim=0)
    return pairs.tolist(), tokens.tolist()


def kernel_bytes(name, T, K, H, n_pairs, n_recv, n_tok_rank, n_local=0, n_local_tok=0):
    """Algorithmic HBM bytes of one launch.  n_tok_rank = distinct (token, destination rank) pairs;
    n_local = received rows whose token lives on this rank (the combine does not move those: the push stores their row number, the
    reduce reads them from the expert output)."""
    row = H + 16
    return {
        "dispatch_stage": T * H * 2 + T * row + n_pairs * 8,    # read bf16 tokens once, write one int8 row per token + the index
        "dispatch_stage_push": T * H * 2 + n_tok_rank * row + n_pairs * 8,   # one row per (token, destination rank) + the index
        # read a token row + index entry per received row, write recv_x / scales / triples; this rank's own tokens (n_local rows of
        # n_local_tok tokens) are read once per TOKEN
        "dispatch_pull": 2 * (n_recv - n_local) * row + (n_recv - n_local) * 8 + (n_local_tok + n_local) * row,
        "combine_push": 2 * (n_recv - n_local) * H * 2 + n_recv * 12 + n_local * 4,   # rows read and written into the owners' slots
        "combine_reduce": n_pairs * H * 2 + T * H * 2,          # read K slots per token, write one bf16 row
    }[name]


def xgmi_projection(pairs_to, tokens_to, kernels_us, small_us, hidden, ep=8, me=0):
    """What an EP = `ep` rank should reach on xGMI, stated BEFORE the first multi-GPU run (N = 1 line): this rank's routing as if its
    256 experts were spread over `ep` ranks (pairs_to / tokens_to from routing_stats(topk_idx, ep, me); by symmetry the rows it
    receives from peer s ~ the pairs it sends to s), the cross-GPU bytes of each leg (push dispatch = one
    (H+16)-byte row per (token, destination rank) + 8 B per pair; combine = one 2H-byte row per remote selection), the time those
    bytes need on (ep-1) links of 153 GB/s -- and on the busiest single link --, and the HBM-side kernel times of `ep8_proxy`
    (every row taking the remote-row code path on this GPU).  Per leg the projection is max(HBM-side kernel, busiest link);
    the step is the sum of the legs + the kernels that stay local + the small launches.  Pure arithmetic on host integers."""
    row = hidden + 16
    peers = [d for d in range(ep) if d != me]
    disp_link = [tokens_to[d] * row + pairs_to[d] * 8 for d in peers]
    comb_link = [pairs_to[d] * hidden * 2 for d in peers]
    peak = XGMI_LINK_GBPS * (ep - 1)
    us = lambda b, gbps: b / gbps / 1e3
    legs = {}
    for name, link, kern in (("dispatch_push", disp_link, "dispatch_stage"), ("combine_push", comb_link, "combine_push")):
        k_us = kernels_us.get(kern) or kernels_us.get(kern + "_push") or 0.0
        all_links_us, max_link_us = us(sum(link), peak), us(max(link), XGMI_LINK_GBPS)
        legs[name] = {"cross_gpu_bytes": int(sum(link)), "max_link_bytes": int(max(link)), "all_links_us": all_links_us,
                      "busiest_link_us": max_link_us, "hbm_side_kernel_us": k_us, "projected_us": max(k_us, max_link_us),
                      "bound": "xgmi" if max_link_us > k_us else "hbm"}
    local_us = sum(kernels_us.get(k, 0.0) for k in ("dispatch_pull", "combine_reduce"))
    small = sum(small_us.values())
    step_us = sum(l["projected_us"] for l in legs.values()) + local_us + small
    cross = sum(l["cross_gpu_bytes"] for l in legs.values())
    link_us = sum(l["busiest_link_us"] for l in legs.values())
    n_rows = sum(pairs_to)
    return {"ep": ep, "link_GBps": XGMI_LINK_GBPS, "links": ep - 1, "peak_GBps": peak, "legs": legs,
            "local_kernels_us": local_us, "small_launches_us": small, "projected_step_ms": step_us / 1e3,
            # reference convention (BF16-equivalent received rows, dispatch + combine) per GPU at the projected step
            "projected_value_GBps_per_gpu": 2 * n_rows * hidden * 2 / (step_us * 1e-6) / 1e9,
            # cross-GPU bytes over the time the two link-facing legs are projected to take: the figure north_star's ">= 70 % of per-GPU
            # xGMI peak" is read against; 1.0 would mean both legs run at the busiest link's rate with nothing else in the way
            "projected_xgmi_frac_during_legs": cross / (sum(l["projected_us"] for l in legs.values()) * 1e-6) / 1e9 / peak,
            "link_bound_floor_ms": link_us / 1e3, "target_frac": 0.70,
            "assumes": "push dispatch transport; remote stores keep all 7 links busy concurrently (rows dealt to peers by the "
                       "MI_EP_PUSH_STRIDE deal); HBM-side kernel times from ep8_proxy on this GPU; unmeasured on xGMI"}


def xgmi_measured(world, rank, transport, tokens_to, pairs_to, rows_from, kernel_us, hidden):
    """Cross-GPU bytes of this rank per step, per leg and per link (= per peer: one xGMI link each), over the
    MEASURED duration of the kernel that moves them (HIP events).  Pure arithmetic: CPU-tested in tests/test_bench_cpu.py."""
    row = hidden + 16
    peers = [d for d in range(world) if d != rank]
    if transport == "push":       # sender writes one row per (token, destination rank) + 8 B per pair
        disp_link = [tokens_to[d] * row + pairs_to[d] * 8 for d in peers]
        disp_kernel = "dispatch_stage_push"
    else:                         # receiver reads one row + one index entry per received row
        disp_link = [rows_from[s] * (row + 8) for s in peers]
        disp_kernel = "dispatch_pull"
    t_disp = kernel_us.get(disp_kernel, 0.0) * 1e-6
    comb_link = [rows_from[s] * hidden * 2 for s in peers]      # rows pushed back to their source rank
    t_comb = kernel_us.get("combine_push", 0.0) * 1e-6
    peak = XGMI_LINK_GBPS * (world - 1)
    xg = {"peak_GBps": peak, "link_GBps": XGMI_LINK_GBPS, "links": world - 1, "dispatch_transport": transport,
          "dispatch_kernel": disp_kernel, "dispatch_bytes": sum(disp_link), "combine_bytes": sum(comb_link),
          "dispatch_max_link_bytes": max(disp_link), "combine_max_link_bytes": max(comb_link)}
    if t_disp:
        xg["dispatch_us"] = t_disp * 1e6
        xg["dispatch_GBps"] = sum(disp_link) / t_disp / 1e9
        xg["dispatch_frac"] = xg["dispatch_GBps"] / peak
        xg["dispatch_max_link_frac"] = max(disp_link) / t_disp / 1e9 / XGMI_LINK_GBPS
    if t_comb:
        xg["combine_us"] = t_comb * 1e6
        xg["combine_GBps"] = sum(comb_link) / t_comb / 1e9
        xg["combine_frac"] = xg["combine_GBps"] / peak
        xg["combine_max_link_frac"] = max(comb_link) / t_comb / 1e9 / XGMI_LINK_GBPS
    if t_disp and t_comb:         # both link-facing legs together: the figure north_star's ">= 70 % of per-GPU xGMI peak" is read against
        xg["legs_GBps"] = (sum(disp_link) + sum(comb_link)) / (t_disp + t_comb) / 1e9
        xg["legs_frac"] = xg["legs_GBps"] / peak
    return xg


def xgmi_roofline(xg, projection, hbm_side, timing):
    """The top-level `roofline` of an N > 1 line: the exchange is xGMI-bound, so the dominant kernel is the link-facing leg that takes
    longer (normally combine_push: 2H bytes per selection against H + 16 per (token, rank)), priced against (N - 1) links x 153 GB/s,
    with the busiest link's own fraction and the projection the N = 1 line stated in advance beside it -- measured vs projected in one
    object.  The HBM-side figure of the same step (the N = 1 line's roofline kernel) is kept as `hbm_side`.  None when no leg was timed."""
    legs = [l for l in ("dispatch", "combine") if l + "_GBps" in xg]
    if not legs:
        return None
    leg = max(legs, key=lambda l: xg[l + "_us"])
    r = {"bound": "xgmi", "kernel": "combine_push" if leg == "combine" else xg["dispatch_kernel"], "achieved": xg[leg + "_GBps"],
         "peak": xg["peak_GBps"], "unit": "GB/s", "frac": xg[leg + "_frac"], "max_link_frac": xg[leg + "_max_link_frac"],
         "algorithmic_bytes": xg[leg + "_bytes"], "avg_launch_us": xg[leg + "_us"], "traffic": None, "timing": timing,
         "peak_source": f"{xg['links']} xGMI links x {xg['link_GBps']} GB/s per direction (MI355X_MICROARCH.md)",
         "both_legs": {"achieved": xg.get("legs_GBps"), "frac": xg.get("legs_frac"), "target_frac": 0.70},
         "hbm_side": hbm_side}
    if projection is not None:
        pl = projection["legs"]["combine_push" if leg == "combine" else "dispatch_push"]
        r["projected"] = {"leg_us": pl["projected_us"], "busiest_link_us": pl["busiest_link_us"], "bound": pl["bound"],
                          "frac_during_legs": projection["projected_xgmi_frac_during_legs"],
                          "step_ms": projection["projected_step_ms"], "ep": projection["ep"],
                          "measured_over_projected": xg[leg + "_us"] / pl["projected_us"] if pl["projected_us"] else None}
    return r


# (N = 1: every received row is one of this rank's own tokens, so the whole pull is the token-wise pull_local_kernel)
PMC_KERNEL_NAMES = {"dispatch_stage": "stage_int8_kernel<false, 1>", "dispatch_pull": "pull_local_kernel<false, true>",
                    "combine_push": "combine_push_kernel<false>", "combine_reduce": "combine_reduce_kernel<false, 8, false>"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` measured by `tools/collect_profiles.sh` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this same command, summarised into profiles/r*_pmc_traffic.json with the gfx950 FETCH_SIZE x2 correction).  The collection
    script fails when a kernel named here is missing from the counters, so the file cannot silently go stale; None when no PMC
    summary is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))["kernels"].get(PMC_KERNEL_NAMES.get(kernel, kernel))
        # (the MLA merge launch: the one behind a headline-size decode launch, not the largest of the run -- tools/summarize_prof.py)
        return k.get("hbm_bytes_per_launch_behind_headline_decode", k["hbm_bytes_per_launch"]) if k else None
    except Exception:  # noqa: BLE001
        return None


def pmc_traffic_source():
    """Which committed counter summary `traffic` values come from (they are read from the file, not collected in this run)."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return os.path.relpath(files[-1], ROOT) if files else None


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only): bounded samples of the same workload on the host cores
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(sample_tokens):
    """(1) `kind: port`, all host cores: the reference's alltoall strategy (normal_strategy.py:481-790) restated with torch CPU
    ops, 8 gloo ranks x (cores / 8) threads, whole passes of the C2 workload (oracle/cpu_alltoall.py);
    (2) `single_core`: the NumPy oracle (oracle/ep.py) on one core, W = 1."""
    import numpy as np

    from oracle import cpu_alltoall as CA
    from oracle import ep as O
    from oracle.bf16 import f32_to_bf16_bits_rne

    cores = os.cpu_count() or 8
    out = {}
    try:
        r = CA.timed_run(W=8, T=sample_tokens, H=HIDDEN, K=TOPK, E=EXPERTS, cores=cores, min_seconds=10.0, max_passes=8)
        bytes_ = 2 * r["rows_all_ranks"] * HIDDEN * 2 * r["passes"]
        out = {"value": bytes_ / r["seconds"] / 1e9, "unit": "GB/s", "cores": r["cores"], "kind": "port",
               "sample": f"{r['passes']} passes of EP=8 x {sample_tokens} tok/rank x hidden {HIDDEN} x top-{TOPK} of {EXPERTS}: layout + "
                         f"int8 dispatch + cast-back + bf16 combine, reference alltoall strategy restated with torch CPU ops over "
                         f"8 gloo ranks x {r['threads_per_rank']} threads (oracle/cpu_alltoall.py), {r['seconds']:.2f} s",
               "seconds": r["seconds"]}
    except Exception as e:  # noqa: BLE001
        out = {"error": f"gloo alltoall baseline failed: {e}"}
    rng = np.random.default_rng(0)
    T1 = min(sample_tokens, 1024)
    x = f32_to_bf16_bits_rne(rng.standard_normal((T1, HIDDEN)).astype(np.float32))
    scores = np.abs(rng.standard_normal((T1, EXPERTS))) + 1
    idx = np.argpartition(-scores, TOPK, axis=1)[:, :TOPK].astype(np.int64)
    w = rng.standard_normal((T1, TOPK)).astype(np.float32)
    t0 = time.perf_counter()
    reps = 0
    while True:
        res = O.normal_dispatch([x], [idx], EXPERTS, quant=True)[0]
        y = O.per_token_cast_back(res.recv_x, res.recv_x_scales)
        O.combine([y], [res.recv_src_idx], [res.total_recv], [idx], [w], EXPERTS)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 5.0 or reps >= 64:
            break
    out["single_core"] = {"value": 2 * res.total_recv * HIDDEN * 2 * reps / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
                          "sample": f"{reps} passes of W=1, {T1} tokens through oracle/ep.py (NumPy), {dt:.2f} s"}
    if "value" not in out:          # the multi-process leg failed: fall back to the single-core number as the baseline
        out.update({k: v for k, v in out["single_core"].items()})
    return out


def mla_cpu_baseline(n_seq=8):
    """The CPU restatement of the reference kernel (oracle/kernels.py decode_mla: per-page online softmax, the arithmetic the
    parity tests check against) on `n_seq` sequences of the C4 shape, torch CPU.  The per-page operands are small (128 x 576 by
    576 x 64), so more than ~16 threads only add synchronisation cost: cores = the threads actually used."""
    from oracle import kernels as OK

    B, Hq, S, page = n_seq, 128, 4096, 64
    g = torch.Generator().manual_seed(0)
    maxp = S // page
    q = torch.randn((B, Hq, 576), generator=g).to(torch.bfloat16)
    kn = torch.randn((B * maxp, page, 1, 512), generator=g).to(torch.bfloat16)
    kr = torch.randn((B * maxp, page, 1, 64), generator=g).to(torch.bfloat16)
    bt = torch.randperm(B * maxp, generator=g).to(torch.int32).reshape(B, maxp)
    lens = torch.full((B,), S, dtype=torch.int32)
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    OK.decode_mla(q[:1], kn, kr, lens[:1], bt[:1], 576 ** -0.5)
    t0 = time.perf_counter()
    reps = 0
    while True:
        OK.decode_mla(q, kn, kr, lens, bt, 576 ** -0.5)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 8.0 or reps >= 50:
            break
    return {"value": B * reps / dt, "unit": "tok/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {B} sequences of 4096 keys x 128 heads x 576/512 through oracle/kernels.py decode_mla (torch CPU), {dt:.2f} s"}


def mla_section(args):
    try:
        from sgl_kernel_npu.bench_hooks import bench_mla_decode     # present once the MLA kernel is built
    except Exception:
        return None
    try:
        r = bench_mla_decode(steps=100, warmup=300)      # MFMA-heavy: let the clocks settle (tens of ms)
        parts = [pmc_traffic(k) for k in r.pop("pmc_kernels", [])]
        if parts and all(v is not None for v in parts):
            r["roofline"]["traffic"] = sum(parts)      # all launches of one decode step
            r["roofline"]["traffic_source"] = pmc_traffic_source()
        if not args.no_cpu_baseline:
            r["cpu_baseline"] = mla_cpu_baseline()
        return r
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


# ---------------------------------------------------------------------------------------------------------------------
# C3 / C5 sections (every rank runs them; rank 0 reports max-over-ranks numbers)
# ---------------------------------------------------------------------------------------------------------------------
def _phases(phases):
    """Run the section's phases in order.  Every rank executes the SAME sequence of host collectives (a barrier after each phase,
    one all-reduce at the end) whether or not its own phase raised: a rank that failed alone must not leave the others waiting
    inside a collective it never enters.  -> (results dict, error string or None)."""
    res, err = {}, None
    for ph in phases:
        if err is None:
            try:
                res.update(ph() or {})
            except Exception as e:  # noqa: BLE001
                err = str(e)[:300]
        try:
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = err or str(e)[:300]
        dist.barrier()
    bad = torch.tensor([0.0 if err is None else 1.0], device="cuda")
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)
    if bad.item() > 0 and err is None:
        err = "another rank failed in this section"
    return res, err


def low_latency_section(buf, rank, world):
    """BASELINE C3: low-latency dispatch + combine, 128 tokens per rank, hidden 7168, top-8, 32 local experts per rank."""
    T, E = 128, 32 * world
    g = torch.Generator(device="cuda").manual_seed(77 + rank)
    x = torch.randn((T, HIDDEN), generator=g, device="cuda").to(torch.bfloat16)
    idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), TOPK, dim=-1)[1]
    w = torch.rand((T, TOPK), generator=g, device="cuda")
    st = {}

    def first():
        (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
        st["y"], st["handle"] = (rx.float() * rs[:, None]).to(torch.bfloat16), handle
        out, _, _ = buf.low_latency_combine(st["y"], idx, w, handle)
        return {"bad": 0.0 if check_round_trip(out, x, w) < 3e-3 else 1.0}

    def time_dispatch():
        d = ev_stats(lambda: buf.low_latency_dispatch(x, idx, T, E, use_fp8=True), n=200)
        return {"d50": d["p50_us"], "d99": d["p99_us"]}

    # A combine is timed BEHIND an untimed dispatch, as it runs in a decode step: a combine that directly follows another combine takes
    # the three-launch form (its two-launch form is only safe behind a dispatch's all-to-all count exchange, deep_ep.hpp), which is what a
    # loop of lone combines would time.  (The lone captured combine of graph_combine below IS that three-launch form.)
    def redispatch():
        _, _, st["handle"], _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)

    def time_combine():
        c = ev_stats(lambda: buf.low_latency_combine(st["y"], idx, w, st["handle"]), n=200, pre=redispatch)
        return {"c50": c["p50_us"], "c99": c["p99_us"]}

    def queued():               # the same calls queued back to back (no host synchronisation between calls)
        d = queued_stats(lambda: buf.low_latency_dispatch(x, idx, T, E, use_fp8=True))
        c = queued_stats(lambda: buf.low_latency_combine(st["y"], idx, w, st["handle"]), pre=redispatch)
        return {"qd50": d["p50_us"], "qd99": d["p99_us"], "qc50": c["p50_us"], "qc99": c["p99_us"],
                "qdh50": d["host_enqueue_us_p50"], "qdhmax": d["host_enqueue_us_max"], "qdhslow": d["host_enqueue_us_of_slowest_sample"],
                "qdslowi": d["slowest_sample_index"]}

    def graph_dispatch():       # the same calls replayed from a captured HIP graph (device-resident epochs make them capturable)
        d = graph_stats(lambda: buf.low_latency_dispatch(x, idx, T, E, use_fp8=True))
        return {"gd50": d["p50_us"], "gd99": d["p99_us"]}

    def graph_combine():
        c = graph_stats(lambda: buf.low_latency_combine(st["y"], idx, w, st["handle"]))
        return {"gc50": c["p50_us"], "gc99": c["p99_us"]}

    def graph_pairs():          # ten dispatch + combine pairs in ONE graph: what a captured decode step pays per pair (the single-call
        n_pairs = 10            # graphs above are dominated by the runtime's graph submission, ~10 us before the first kernel starts)
        def run():
            keep = []
            for _ in range(n_pairs):
                (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x, idx, T, E, use_fp8=True)
                out, _, _ = buf.low_latency_combine(st["y"], idx, w, handle)
                keep.append((rx, rs, cnt, out))
            return keep
        p = graph_stats(run, n=50)
        return {"gp50": p["p50_us"] / n_pairs, "gp99": p["p99_us"] / n_pairs}

    res, err = _phases([first, time_dispatch, time_combine, queued, graph_dispatch, graph_combine, graph_pairs])
    if err is not None:
        return {"error": err}
    m = max_over_ranks(res)
    n_sel = T * TOPK
    return {"config": f"low-latency dispatch(int8)+combine(bf16), EP={world}, 128 tok/rank, hidden {HIDDEN}, top-{TOPK} of {E} (BASELINE C3)",
            "dispatch_us_p50": m["d50"], "dispatch_us_p99": m["d99"], "combine_us_p50": m["c50"], "combine_us_p99": m["c99"],
            # each call captured once in a HIP graph and replayed (200 replays): no host work between its kernels
            # 200 calls queued back to back, one synchronisation at the end (the p50 / p99 above synchronise after every call: the GPU idles
            # in between and its clocks sag, which is what a lone decode step sees, not a streaming one)
            "queued": {"dispatch_us_p50": m["qd50"], "dispatch_us_p99": m["qd99"], "combine_us_p50": m["qc50"], "combine_us_p99": m["qc99"],
                       # reading the tail: host time to enqueue one dispatch call (p50 / max) and of the slowest device sample -- when that is
                       # of the order of the p99, the GPU was waiting for the host between the two events, not running a slow kernel
                       "dispatch_host_enqueue_us_p50": m["qdh50"], "dispatch_host_enqueue_us_max": m["qdhmax"],
                       "dispatch_host_enqueue_us_of_slowest_sample": m["qdhslow"], "dispatch_slowest_sample_index": m["qdslowi"]},
            "graph_replay": {"dispatch_us_p50": m["gd50"], "dispatch_us_p99": m["gd99"], "combine_us_p50": m["gc50"], "combine_us_p99": m["gc99"],
                             # per dispatch + combine PAIR inside a graph of ten pairs (submission amortised)
                             "pair_us_p50_in_graph_of_10": m["gp50"], "pair_us_p99_in_graph_of_10": m["gp99"]},
            # reference byte convention (tests/python/deepep/test_low_latency.py:310-322)
            "dispatch_GBps": n_sel * (HIDDEN + HIDDEN // 128 * 4 + 16) / m["d50"] / 1e3, "combine_GBps": n_sel * HIDDEN * 2 / m["c50"] / 1e3,
            # launch forms of the timed dispatch / of a combine behind a dispatch (0 three launches, 2 two launches: deep_ep.hpp)
            "launch_forms": list(buf.runtime.get_low_latency_default_forms()) if hasattr(buf.runtime, "get_low_latency_default_forms") else None,
            "validated_round_trip": m["bad"] == 0.0, "reference_A3_us": {"dispatch": 132, "combine": 126}}


def fused_moe_section(buf, rank, world, T=4096):
    """BASELINE C5: fused_deep_moe, DeepSeek-V3 expert shapes (hidden 7168, 2I = 4096), 32 local experts per rank, T tokens per rank."""
    L = 32
    E = L * world
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fused_f64 as F                      # validation only: per-token evaluation of sampled tokens (tests/fused_f64.py)
    w13_o, w2, s13_o, s2 = F.fused_weights(99 + rank, L, HIDDEN, INTER)          # original column order (gate rows, then up rows)
    perm = F.fusion_perm(2 * INTER)
    w13, s13 = w13_o[:, perm, :].contiguous(), s13_o[:, perm].contiguous()       # the fusion-tile order the op consumes
    del w13_o, s13_o
    g = torch.Generator(device="cuda").manual_seed(199 + rank)
    x = torch.randn((T, HIDDEN), generator=g, device="cuda").to(torch.bfloat16)
    idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), TOPK, dim=-1)[1]
    w = torch.rand((T, TOPK), generator=g, device="cuda")
    f = lambda: buf.fused_deep_moe(x, idx, w, w13, s13, w2, s2, T, E)
    st = {}

    def first():
        out, _ = f()
        torch.cuda.synchronize()
        st["out"] = out.clone()
        return {"finite": 1.0 if bool(torch.isfinite(out.float()).all()) else 0.0}

    def validate():
        # 256 sampled tokens of this rank against the per-token evaluation with exact integer products, at the reference test's bar
        # (tests/python/deepep/test_fused_deep_moe.py:470: avg relative diff < 4e-4).  Runs after the timed phases: it regenerates
        # every owner rank's weights from their seeds (1.4 GB at a time).
        r = F.sampled_check(st.pop("out"), x, idx, w, lambda rr: F.fused_weights(99 + rr, L, HIDDEN, INTER), L, n_samples=256, seed=rank)
        return {"val_ok": 1.0 if r["ok"] else 0.0, "val_avg": r["avg_diff"], "val_calc": r["calc_diff"]}

    def profile():
        buf.begin_profile(0, 10, "")
        for _ in range(10):
            f()
        buf.end_profile()
        st["prof"] = {k: ms / n * 1e3 for k, (n, ms) in buf.get_profile_summary().items() if n}

    def timed():
        r = ev_stats(f, n=20, warm=20)
        return {"p50": r["p50_us"], "p99": r["p99_us"]}

    def clocks():
        # shader clock the chip held under the two grouped GEMMs of the LAST timed call (stamped by the kernels' first workgroup)
        if not hasattr(buf.runtime, "get_gemm_clock"):
            return {}
        c = buf.runtime.get_gemm_clock()
        return {"clk_gemm1": c[0][0], "clk_gemm2": c[2][0] or c[1][0]}

    def vendor_gemm():
        # calibration, not a target: what the vendor library's DENSE int8 GEMM (hipBLASLt through torch._int_mm, int32 output, no
        # epilogue, no grouping) reaches on this chip for GEMM1's shape -- the 3.9 POPS datasheet peak is not attainable in practice
        a = torch.randint(-8, 8, (T * TOPK, HIDDEN), device="cuda", dtype=torch.int32).to(torch.int8)
        wd = w13[0].t()
        r = ev_stats(lambda: torch._int_mm(a, wd), n=5, warm=3)
        return {"vendor": 2.0 * T * TOPK * HIDDEN * 2 * INTER / (r["p50_us"] * 1e-6) / 1e12}

    res, err = _phases([first, profile, timed, clocks, vendor_gemm, validate])
    if err is not None:
        return {"error": err}
    finite = min(res.pop("finite"), res.pop("val_ok"))
    m = max_over_ranks(res)                      # val_avg / val_calc: the worst rank's
    fin = torch.tensor([finite], device="cuda")
    dist.all_reduce(fin, op=dist.ReduceOp.MIN)
    ops = T * TOPK * (HIDDEN * 2 * INTER + INTER * HIDDEN) * 2          # per rank under balanced routing
    tops = ops / (m["p50"] * 1e-6) / 1e12
    return {"config": f"fused_deep_moe, EP={world}, {T} tok/rank, hidden {HIDDEN}, 2I={2 * INTER}, top-{TOPK}, {L} local experts per rank (BASELINE C5)",
            "ms_p50": m["p50"] / 1e3, "ms_p99": m["p99"] / 1e3, "int8_TOPs_per_gpu": tops,
            "roofline": {"bound": "mfma", "achieved": tops, "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": tops / INT8_PEAK_TOPS,
                         "traffic": None},
            # The datasheet peak assumes the 2.4 GHz shader clock; under dense INT8 MFMA issue the chip holds less (power management).
            # Measured inside the two GEMM launches of the last timed call (s_memtime over s_memrealtime, first workgroup): the peak the
            # chip could have delivered AT THAT CLOCK, and the GEMMs' own rate against it (whole-call frac stays against the datasheet).
            "shader_clock_GHz": {"gemm1_swiglu": m.get("clk_gemm1"), "gemm2_push": m.get("clk_gemm2"), "nominal": 2.4},
            "effective_peak_TOPs": (INT8_PEAK_TOPS * min(m["clk_gemm1"], 2.4) / 2.4) if m.get("clk_gemm1") else None,
            "frac_of_effective_peak": (tops / (INT8_PEAK_TOPS * min(m["clk_gemm1"], 2.4) / 2.4)) if m.get("clk_gemm1") else None,
            "vendor_dense_int8_gemm_TOPs": m.get("vendor"),      # hipBLASLt dense GEMM of GEMM1's shape on the same GPU (calibration)
            # ISOLATED, event-timed launches (a second pass with a HIP event pair around every launch: the chain is serialised and every
            # kernel starts behind an event): their sum exceeds the queued call's ms_p50 -- `kernels_sum_over_call` says by how much -- and
            # the per-kernel durations that are comparable with profiles/ are rocprofv3's (profiles/rNN_kernel_stats_by_grid.csv)
            "kernels_avg_us": st.get("prof", {}),
            "kernels_timing": "isolated, event-timed",
            "kernels_sum_over_call": (sum(st.get("prof", {}).values()) / m["p50"]) if st.get("prof") else None,
            # every rank: output finite AND 256 sampled tokens within the reference bar of the per-token evaluation (tests/fused_f64.py)
            "validated": bool(fin.item() > 0), "validation": {"samples_per_rank": 256, "avg_diff_max": m["val_avg"], "calc_diff_max": m["val_calc"],
                                                              "bar": {"avg_diff": 4e-4, "calc_diff": 1e-5}}}


# ---------------------------------------------------------------------------------------------------------------------
def timed_steps(buf, x, topk_idx, topk_w, y, steps, warmup, profiled):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize pairs; max over ranks.  The host collectives are
    executed by every rank whether or not its own steps raised (see _phases); -> (seconds, profile dict, error or None)."""
    err = None
    try:
        for _ in range(warmup):
            one_step(buf, x, topk_idx, topk_w, y)
        if profiled:
            buf.begin_profile(0, steps, "")
        flush_cache()
    except Exception as e:  # noqa: BLE001
        err = str(e)[:300]
    barrier_sync()
    t0 = time.perf_counter()
    if err is None:
        try:
            for _ in range(steps):
                one_step(buf, x, topk_idx, topk_w, y)
        except Exception as e:  # noqa: BLE001
            err = str(e)[:300]
    try:
        barrier_sync()
    except Exception as e:  # noqa: BLE001
        err = err or str(e)[:300]
    dt = time.perf_counter() - t0
    prof = {}
    if profiled:
        try:
            buf.end_profile()
            prof = buf.get_profile_summary()
        except Exception as e:  # noqa: BLE001
            err = err or str(e)[:300]
    tmax = torch.tensor([dt, 0.0 if err is None else 1.0], device="cuda", dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if tmax[1].item() > 0 and err is None:
        err = "another rank failed"
    return float(tmax[0].item()), prof, err


def main():
    args = parse()
    relaunch = self_launch_command(args, os.environ)
    if relaunch is not None:
        os.execvpe(relaunch[0][0], relaunch[0], relaunch[1])
    rank, world = init_dist(args.gpus)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    import deep_ep

    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "10000")     # bounded spins: a broken peer mapping fails fast, then falls back
    T = args.tokens
    x, topk_idx, topk_w = make_inputs(rank, T)
    group = dist.group.WORLD
    strategy = args.strategy
    buf = deep_ep.Buffer(group, low_latency_mode=True, normal_strategy=strategy, low_latency_strategy=strategy)
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
        buf = deep_ep.Buffer(group, low_latency_mode=True, normal_strategy="alltoall", low_latency_strategy="alltoall")
        out, n_recv, y, recv, handle = one_step(buf, x, topk_idx, topk_w, None)
        torch.cuda.synchronize()
        validated = check_round_trip(out, x, topk_w) < 3e-3
    strategy = buf.normal_strategy.get_name()
    windowed = strategy == "default" and hasattr(buf.runtime, "get_profile_summary")

    # ---- the timed region(s): exactly K steps per transport, barrier + synchronize on both sides, max over ranks
    transports = [None]
    if windowed and hasattr(buf.runtime, "set_dispatch_transport"):
        transports = ["push", "pull"] if world > 1 else [buf.runtime.get_dispatch_transport()]
    runs = {}
    for tr in transports:
        if tr is not None:
            buf.runtime.set_dispatch_transport(tr)
        # the timed region: exactly K steps, no instrumentation.  Kernel durations (roofline) come from a SECOND pass of the same
        # K steps with a HIP event pair around every kernel: each pair costs ~10 us of bubble at a kernel boundary (0.396 -> 0.448 ms
        # per step measured), which must not be charged to the hot path; that pass's own time is reported as profiled_ms_per_step.
        dt, _, err = timed_steps(buf, x, topk_idx, topk_w, y, args.steps, args.warmup, False)
        prof, dt_prof = {}, None
        if err is None and windowed:
            dt_prof, prof, err = timed_steps(buf, x, topk_idx, topk_w, y, args.steps, 0, True)
        if err is None:
            runs[tr] = (dt, prof, dt_prof)
        elif rank == 0:
            print(f"[bench] transport {tr} failed: {err}", file=sys.stderr)
    assert runs, "no dispatch transport completed"
    best = min(runs, key=lambda k: runs[k][0])
    dt, prof, dt_prof = runs[best]
    if best is not None:
        buf.runtime.set_dispatch_transport(best)
    rows = torch.tensor([n_recv], device="cuda", dtype=torch.float64)
    dist.all_reduce(rows, op=dist.ReduceOp.SUM)
    total_rows = float(rows.item())
    ms_per_step = dt / args.steps * 1e3
    bytes_per_step = 2 * total_rows * HIDDEN * 2          # dispatch recv + combine send, BF16-equivalent (reference convention)
    value = bytes_per_step / (ms_per_step * 1e-3) / 1e9

    # ---- the same K steps without the host on the critical path (dispatch(num_worst_tokens = T * K * W): worst-case sized outputs, the
    # host never reads the receive count): what a serving stack that captures the step in a graph pays
    nosync = None
    if windowed:
        try:
            worst = T * TOPK * world
            out_ns, y_ns = one_step_nosync(buf, x, topk_idx, topk_w, None, worst)
            torch.cuda.synchronize()
            ok_ns = check_round_trip(out_ns, x, topk_w) < 3e-3
            err_ns = None
        except Exception as e:  # noqa: BLE001
            ok_ns, err_ns, y_ns, worst = False, str(e)[:300], None, 0
        for _ in range(2):
            if err_ns is None:
                try:
                    one_step_nosync(buf, x, topk_idx, topk_w, y_ns, worst)
                except Exception as e:  # noqa: BLE001
                    err_ns = str(e)[:300]
        flush_cache()
        barrier_sync()
        t0 = time.perf_counter()
        if err_ns is None:
            try:
                for _ in range(args.steps):
                    one_step_nosync(buf, x, topk_idx, topk_w, y_ns, worst)
            except Exception as e:  # noqa: BLE001
                err_ns = str(e)[:300]
        barrier_sync()
        tns = torch.tensor([time.perf_counter() - t0, 0.0 if (err_ns is None and ok_ns) else 1.0], device="cuda", dtype=torch.float64)
        dist.all_reduce(tns, op=dist.ReduceOp.MAX)
        nosync = {"error": err_ns or "round trip check failed on some rank"} if tns[1].item() > 0 else \
            {"ms_per_step": float(tns[0].item()) / args.steps * 1e3, "num_worst_tokens": worst}
        del y_ns

    # ---- EP = 8 proxy on whatever this run has: with the own-rank shortcuts off EVERY row takes the path a remote row takes (dispatch:
    # pull_indexed / stage_push for all rows; combine: every row pushed through the window), which is what 7/8 of the rows of an EP = 8
    # rank do.  Same results; the kernel times transfer to EP = 8 as HBM-side costs (the xGMI legs come on top).
    proxy = None
    if windowed and hasattr(buf.runtime, "set_local_row_paths"):
        was = list(buf.runtime.get_local_row_paths())
        buf.runtime.set_local_row_paths(False, False)
        try:
            dt_p, _, err_p = timed_steps(buf, x, topk_idx, topk_w, y, args.steps, 2, False)
            dt_pp, prof_p, err_pp = timed_steps(buf, x, topk_idx, topk_w, y, args.steps, 0, True) if err_p is None else (None, {}, err_p)
            proxy = {"error": err_p or err_pp} if (err_p or err_pp) else {"ms_per_step": dt_p / args.steps * 1e3, "prof": prof_p}
        finally:
            buf.runtime.set_local_row_paths(*was)

    # ---- the other BASELINE configs, same process group (every rank takes part)
    extra = {}
    if windowed and not args.no_extra:
        for name, fn in (("low_latency", low_latency_section), ("fused_deep_moe", fused_moe_section)):
            extra[name] = fn(buf, rank, world)

    pairs_to, tokens_to = routing_stats(topk_idx, world, rank)
    rows_from = torch.bincount(handle[3][:3 * n_recv].view(-1, 3)[:, 0].long(), minlength=world).tolist() if windowed else [0] * world
    if rank != 0:
        dist.destroy_process_group()
        flush_c_stdout()
        return
    n_pairs = int(sum(pairs_to))
    n_tok_rank = int(sum(tokens_to))
    result = {
        "metric": "dispatch+combine GB/s (EP=N, 4096 tok/rank, h=7168, top-8, INT8 dispatch / BF16 combine; "
                  "reference convention: BF16-equivalent received rows / time)",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int8/bf16", "data": "synthetic",
        "config": {"workload": f"deep_ep normal dispatch(int8)+combine(bf16), EP={world}, {T} tok/rank, hidden {HIDDEN}, "
                               f"top-{TOPK} of {EXPERTS} experts (BASELINE C2 shapes at EP={world})",
                   "strategy": strategy, "dispatch_transport": best, "tokens_per_rank": T, "hidden": HIDDEN, "topk": TOPK,
                   "experts": EXPERTS, "cache_flush": "256 MB write before the timed region"},
        "per_gpu_GBps": value / world, "validated_round_trip": bool(validated),
        "profiled_ms_per_step": dt_prof / args.steps * 1e3 if dt_prof else None,     # second pass of K steps with HIP events (roofline)
    }
    if len(runs) > 1:
        result["transports"] = {k: {"ms_per_step": v[0] / args.steps * 1e3, "value": bytes_per_step / (v[0] / args.steps) / 1e9}
                                for k, v in runs.items()}
    if prof:
        per = {k: {"launches": n, "avg_us": ms / n * 1e3} for k, (n, ms) in prof.items() if n}
        bulk = [k for k in per if k in ("dispatch_stage", "dispatch_stage_push", "dispatch_pull", "combine_push", "combine_reduce")]
        dom = max(bulk, key=lambda k: per[k]["avg_us"])
        local_paths = list(buf.runtime.get_local_row_paths()) if hasattr(buf.runtime, "get_local_row_paths") else [True, True]
        n_local = rows_from[rank] if windowed and local_paths[1] else 0
        n_local_d = rows_from[rank] if windowed and local_paths[0] else 0
        kb = lambda k: kernel_bytes(k, T, TOPK, HIDDEN, n_pairs, n_recv, n_tok_rank, n_local if k.startswith("combine") else n_local_d,
                                    tokens_to[rank] if n_local_d else 0)
        alg = kb(dom)
        achieved = alg / (per[dom]["avg_us"] * 1e-6) / 1e9
        result["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "timing": f"HIP event pairs on the kernels' stream, {args.steps} launches, pass right after the timed region",
                              "frac": achieved / HBM_PEAK_GBPS, "traffic": pmc_traffic(dom) if world == 1 else None,
                              "traffic_source": pmc_traffic_source() if world == 1 else None,
                              "algorithmic_bytes": alg, "avg_launch_us": per[dom]["avg_us"]}
        result["kernels"] = {k: dict(per[k], GBps=kb(k) / (per[k]["avg_us"] * 1e-6) / 1e9) for k in bulk}
        result["kernels"].update({k: per[k] for k in per if k not in bulk})
        if world > 1:
            result["xgmi"] = xgmi_measured(world, rank, best, tokens_to, pairs_to, rows_from, {k: v["avg_us"] for k, v in per.items()}, HIDDEN)
    if proxy is not None:
        if "prof" in proxy:
            perp = {k: ms / n * 1e3 for k, (n, ms) in proxy.pop("prof").items() if n}
            kbp = lambda k: kernel_bytes(k, T, TOPK, HIDDEN, n_pairs, n_recv, n_tok_rank, 0, 0)
            proxy["kernels"] = {k: {"avg_us": perp[k], "GBps": kbp(k) / (perp[k] * 1e-6) / 1e9, "algorithmic_bytes": kbp(k)}
                                for k in ("dispatch_stage", "dispatch_stage_push", "dispatch_pull", "combine_push", "combine_reduce") if k in perp}
            proxy["small_launches_us"] = {k: v for k, v in perp.items() if k not in proxy["kernels"]}
            proxy["value"] = bytes_per_step / (proxy["ms_per_step"] * 1e-3) / 1e9
            proxy["what"] = ("own-rank shortcuts off (set_local_row_paths(False, False)): every row staged, pulled by index and pushed "
                             "through the window like a remote row; HBM-side kernel times of an EP = 8 rank, xGMI legs not included")
        result["ep8_proxy"] = proxy
    if nosync is not None:
        if "ms_per_step" in nosync:
            nosync["value"] = bytes_per_step / (nosync["ms_per_step"] * 1e-3) / 1e9
            nosync["what"] = ("same K steps through dispatch(num_worst_tokens=T*K*W): worst-case sized outputs, no host read of the "
                              "receive count between dispatch and combine (the headline's dispatch reads it: one pinned-word spin per step)")
        result["no_host_sync"] = nosync
    if proxy is not None and "kernels" in proxy:
        # N = 1: this rank's routing as if its experts were spread over 8 ranks, stated in advance; N > 1: the run's own routing
        ep, me = (8, 0) if world == 1 else (world, rank)
        p8, t8 = routing_stats(topk_idx, ep, me)
        result["xgmi_projection"] = xgmi_projection(p8, t8, {k: v["avg_us"] for k, v in proxy["kernels"].items()},
                                                    proxy.get("small_launches_us", {}), HIDDEN, ep=ep, me=me)
    if world > 1 and "xgmi" in result and "roofline" in result:
        xr = xgmi_roofline(result["xgmi"], result.get("xgmi_projection"), result["roofline"], result["roofline"]["timing"])
        if xr is not None:
            result["roofline"] = xr
    if args.dry_run_8:
        result["dry_run_single_device"] = True       # every rank ran on cuda:0: plumbing only, not a measurement
    result.update(extra)
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
