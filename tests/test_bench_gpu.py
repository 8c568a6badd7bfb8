"""bench.py contract: the launch line the driver uses for N > 1, exercised on ONE GPU (every rank on cuda:0, gloo bootstrap --
BENCH_SINGLE_DEVICE=1), and the default N = 1 run.  The JSON object must be the LAST line of stdout."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _last_json(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.strip()]
    assert lines, "bench.py printed nothing"
    return json.loads(lines[-1])          # must parse: nothing may follow the JSON line


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # plain `python bench.py --gpus 2`, no RANK in the environment: bench.py re-executes itself through torch.distributed.run (the
    # launch line the driver uses for N > 1), so this one call covers the self-launch AND the launcher form
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["strategy"] == "default" and d["validated_round_trip"] is True
    # N > 1: the top-level roofline is the link-facing leg (the HBM-side kernel of the N = 1 line rides along as `hbm_side`)
    rf = d["roofline"]
    assert rf["bound"] == "xgmi" and rf["unit"] == "GB/s" and rf["peak"] > 0 and rf["achieved"] > 0 and 0 < rf["frac"]
    assert rf["kernel"] in ("combine_push", d["xgmi"]["dispatch_kernel"]) and rf["hbm_side"]["bound"] == "hbm"
    # N > 1: both dispatch transports timed with the same K steps, the cross-GPU legs priced per leg and per link, C3 / C5 emitted
    assert set(d["transports"]) == {"push", "pull"} and d["config"]["dispatch_transport"] in ("push", "pull")
    assert {"dispatch_frac", "combine_frac", "dispatch_max_link_bytes", "combine_max_link_bytes"} <= set(d["xgmi"])
    assert d["low_latency"]["validated_round_trip"] is True and d["low_latency"]["dispatch_us_p50"] > 0
    assert d["fused_deep_moe"].get("validated") is True and d["fused_deep_moe"]["validation"]["avg_diff_max"] < 4e-4, d["fused_deep_moe"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-mla"], cwd=ROOT,
                       capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in REQUIRED + ("roofline", "cpu_baseline", "low_latency", "fused_deep_moe"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["dtype"] == "int8/bf16" and d["vs_baseline"] is None
    assert d["cpu_baseline"]["cores"] >= 1 and "single_core" in d["cpu_baseline"]
    assert d["fused_deep_moe"]["roofline"]["bound"] == "mfma" and d["fused_deep_moe"]["validated"] is True
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source")) <= set(d["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    # the number that transfers to EP = 8: every row through the window paths, kernel by kernel
    px = d["ep8_proxy"]
    assert px["ms_per_step"] > d["ms_per_step"] and {"dispatch_pull", "combine_push", "combine_reduce"} <= set(px["kernels"])
    assert px["kernels"]["combine_push"]["algorithmic_bytes"] > 9e8
    # the stated expectation for the first EP = 8 run: per-leg cross-GPU bytes, link-bound time, projected step
    xp = d["xgmi_projection"]
    assert xp["ep"] == 8 and xp["peak_GBps"] == 7 * 153.0 and set(xp["legs"]) == {"dispatch_push", "combine_push"}
    assert xp["legs"]["combine_push"]["cross_gpu_bytes"] > 3.5e8 and xp["projected_step_ms"] > xp["link_bound_floor_ms"] > 0.3


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_dry_run_eight_ranks_one_gpu():
    """`bench.py --dry-run-8`: the 8-GPU launch line on ONE GPU at full C2 size (every rank on cuda:0, hipIpc windows): every key a real
    8-GPU line carries must be there -- both transports, per-leg and per-link xGMI pricing, C3 and C5 sections validated."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-8", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["dry_run_single_device"] is True and d["scaling"] == "weak" and d["validated_round_trip"] is True
    assert d["config"]["strategy"] == "default" and d["config"]["tokens_per_rank"] == 4096
    assert set(d["transports"]) == {"push", "pull"}
    xg = d["xgmi"]
    for k in ("peak_GBps", "link_GBps", "dispatch_bytes", "combine_bytes", "dispatch_max_link_bytes", "combine_max_link_bytes",
              "dispatch_GBps", "dispatch_frac", "dispatch_max_link_frac", "combine_GBps", "combine_frac", "combine_max_link_frac"):
        assert k in xg and xg[k] > 0, k
    assert xg["peak_GBps"] == 7 * 153.0
    assert d["low_latency"]["validated_round_trip"] is True and d["fused_deep_moe"]["validated"] is True
    assert "ep8_proxy" in d and "kernels" in d["ep8_proxy"]
