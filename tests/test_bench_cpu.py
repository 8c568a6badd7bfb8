"""bench.py host logic that needs no GPU: the self-launch decision for --gpus N > 1 and the xGMI projection arithmetic."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(**kw):
    d = dict(gpus=1, steps=30, warmup=5, tokens=4096, strategy="default", no_cpu_baseline=False, no_mla=False, no_extra=False,
             dry_run_8=False)
    d.update(kw)
    return argparse.Namespace(**d)


def test_self_launch_decision():
    B = _bench()
    # N = 1, or already a rank of some launcher: run in this process
    assert B.self_launch_command(_args(gpus=1), {}) is None
    assert B.self_launch_command(_args(gpus=8), {"RANK": "3", "WORLD_SIZE": "8"}) is None
    assert B.self_launch_command(_args(dry_run_8=True), {"RANK": "0"}) is None
    # plain `python bench.py --gpus 8`: re-executed through torch.distributed.run, one rank per GPU, flags carried over
    cmd, env = B.self_launch_command(_args(gpus=8, steps=7, warmup=2, no_extra=True), {"PATH": "/usr/bin"})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail[:6] == ["--gpus", "8", "--steps", "7", "--warmup", "2"] and "--no-extra" in tail and "--dry-run-8" not in tail
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "BENCH_SINGLE_DEVICE" not in env and env["PATH"] == "/usr/bin"
    cmd, env = B.self_launch_command(_args(gpus=2), {})
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--gpus") + 1] == "2"
    # --dry-run-8: the same re-execution with every rank on cuda:0
    cmd, env = B.self_launch_command(_args(dry_run_8=True), {})
    assert env["BENCH_SINGLE_DEVICE"] == "1" and "--dry-run-8" in cmd and cmd[cmd.index("--gpus") + 1] == "8"
    # the re-executed ranks parse what the parent passed
    sys_argv = sys.argv
    try:
        sys.argv = ["bench.py"] + tail
        a = B.parse()
        assert a.gpus == 8 and a.steps == 7 and a.warmup == 2 and a.no_extra
    finally:
        sys.argv = sys_argv


def test_xgmi_projection_arithmetic():
    B = _bench()
    H = 7168
    # balanced top-8 of 256 over 8 ranks: 4096 pairs to every rank, ~2687 distinct tokens per destination
    pairs, toks = [4096] * 8, [2687] * 8
    kern = {"dispatch_stage": 17.0, "dispatch_pull": 73.0, "combine_push": 154.0, "combine_reduce": 86.0}
    p = B.xgmi_projection(pairs, toks, kern, {"layout": 10.0, "dispatch_notify": 9.0}, H)
    d, c = p["legs"]["dispatch_push"], p["legs"]["combine_push"]
    assert d["cross_gpu_bytes"] == 7 * (2687 * (H + 16) + 4096 * 8) and c["cross_gpu_bytes"] == 7 * 4096 * 2 * H
    assert abs(d["busiest_link_us"] - (2687 * (H + 16) + 4096 * 8) / 153e3) < 1e-9
    assert abs(c["busiest_link_us"] - 4096 * 2 * H / 153e3) < 1e-9
    assert d["bound"] == "xgmi" and c["bound"] == "xgmi" and c["projected_us"] == c["busiest_link_us"]
    want = d["busiest_link_us"] + c["busiest_link_us"] + 73.0 + 86.0 + 19.0
    assert abs(p["projected_step_ms"] * 1e3 - want) < 1e-6
    assert abs(p["projected_xgmi_frac_during_legs"] - 1.0) < 1e-9 and p["peak_GBps"] == 7 * 153.0
    # a skewed routing: the busiest link sets the leg, the fraction falls below 1
    pairs2 = [4096, 8192, 2048] + [4096] * 5
    p2 = B.xgmi_projection(pairs2, toks, kern, {}, H)
    assert p2["legs"]["combine_push"]["busiest_link_us"] == 8192 * 2 * H / 153e3 and p2["projected_xgmi_frac_during_legs"] < 1.0
    # a leg whose HBM-side kernel is slower than its link time is HBM-bound
    p3 = B.xgmi_projection(pairs, toks, dict(kern, combine_push=900.0), {}, H)
    assert p3["legs"]["combine_push"]["bound"] == "hbm" and p3["legs"]["combine_push"]["projected_us"] == 900.0


def test_multi_gpu_line_names_the_xgmi_leg():
    """At world > 1 the top-level roofline is the link-facing leg (bound "xgmi", peak = 7 x 153 GB/s), measured beside projected; reference
    bandwidth convention: tests/python/deepep/test_intranode.py:447-448,530-534 (bytes moved over the kernel's time)."""
    B = _bench()
    H, W = 7168, 8
    pairs, toks = [4096] * W, [2687] * W
    rows_from = [4096] * W
    kern_us = {"dispatch_stage_push": 150.0, "dispatch_pull": 73.0, "combine_push": 420.0, "combine_reduce": 86.0, "layout": 10.0}
    xg = B.xgmi_measured(W, 3, "push", toks, pairs, rows_from, kern_us, H)
    assert xg["peak_GBps"] == 7 * 153.0 and xg["links"] == 7
    assert xg["dispatch_bytes"] == 7 * (2687 * (H + 16) + 4096 * 8) and xg["combine_bytes"] == 7 * 4096 * 2 * H
    assert abs(xg["combine_GBps"] - xg["combine_bytes"] / 420e-6 / 1e9) < 1e-6
    assert abs(xg["combine_max_link_frac"] - 4096 * 2 * H / 420e-6 / 1e9 / 153.0) < 1e-9
    assert abs(xg["legs_GBps"] - (xg["dispatch_bytes"] + xg["combine_bytes"]) / 570e-6 / 1e9) < 1e-6
    proj = B.xgmi_projection(pairs, toks, {"dispatch_stage": 17.0, "dispatch_pull": 73.0, "combine_push": 154.0, "combine_reduce": 86.0},
                             {}, H, ep=W, me=3)
    hbm = {"bound": "hbm", "kernel": "combine_reduce", "frac": 0.7, "timing": "t"}
    r = B.xgmi_roofline(xg, proj, hbm, "t")
    assert r["bound"] == "xgmi" and r["kernel"] == "combine_push" and r["peak"] == 7 * 153.0 and r["unit"] == "GB/s"
    assert r["achieved"] == xg["combine_GBps"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["max_link_frac"] == xg["combine_max_link_frac"] and r["hbm_side"] is hbm and r["traffic"] is None
    assert r["both_legs"]["frac"] == xg["legs_frac"] and r["both_legs"]["target_frac"] == 0.70
    pl = proj["legs"]["combine_push"]
    assert r["projected"]["leg_us"] == pl["projected_us"] and abs(r["projected"]["measured_over_projected"] - 420.0 / pl["projected_us"]) < 1e-12
    # the pull transport prices the receiver's reads; a dispatch leg slower than the combine leg becomes the roofline kernel
    xg2 = B.xgmi_measured(W, 0, "pull", toks, pairs, rows_from, dict(kern_us, dispatch_pull=900.0), H)
    assert xg2["dispatch_kernel"] == "dispatch_pull" and xg2["dispatch_bytes"] == 7 * 4096 * (H + 16 + 8)
    r2 = B.xgmi_roofline(xg2, None, hbm, "t")
    assert r2["kernel"] == "dispatch_pull" and "projected" not in r2
    # no timed leg: no xGMI roofline (the caller keeps the HBM one)
    assert B.xgmi_roofline(B.xgmi_measured(W, 0, "push", toks, pairs, rows_from, {}, H), proj, hbm, "t") is None
    # W = 2: one link
    xg3 = B.xgmi_measured(2, 1, "push", [100, 50], [300, 200], [250, 222], {"combine_push": 10.0, "dispatch_stage_push": 5.0}, H)
    assert xg3["peak_GBps"] == 153.0 and xg3["combine_bytes"] == 250 * 2 * H and xg3["dispatch_bytes"] == 100 * (H + 16) + 300 * 8
