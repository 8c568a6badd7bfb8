"""Turn rocprofv3 CSV output (gpurun_out/...) into the small tracked summaries under profiles/.

  python tools/summarize_prof.py <round> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]

Writes profiles/r<round>_kernel_stats.csv (top kernels of `--kernel-trace --stats`) and, when PMC passes are given,
profiles/r<round>_pmc_traffic.json: per kernel, per launch, HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by exactly 2x --
/opt/skills/guides/MI355X_MICROARCH.md section HBM -- and WRITE_SIZE matched a known byte count 1:1 in these kernels)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(mi_[a-z_]+::)?([A-Za-z0-9_]+)(<[^(]*>)?\(", name)
    return (m.group(2) + (m.group(3) or "")) if m else name[:60]


def by_grid(rnd, stats_dir, out_dir):
    """profiles/r<round>_kernel_stats_by_grid.csv: one row per (kernel, grid size, workgroup size) from the per-dispatch kernel trace --
    the same kernel runs at several problem sizes inside one bench (C2-size, decode-size, validation launches), and an average over
    all of them prices nothing.  `frac` of a bench object can be recomputed from this file alone: algorithmic bytes
    / avg_ns of the row with the largest grid of that kernel."""
    files = glob.glob(os.path.join(stats_dir, "*kernel_trace.csv"))
    if not files:
        return
    agg = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        try:
            grid = tuple(int(r.get(f"Grid_Size_{a}", r.get(f"Grid_Size{a}", 0)) or 0) for a in "XYZ")
            wg = tuple(int(r.get(f"Workgroup_Size_{a}", r.get(f"Workgroup_Size{a}", 0)) or 0) for a in "XYZ")
            agg[(short(r["Kernel_Name"]), grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        except (KeyError, ValueError):
            continue
    ours = ("stage", "pull", "combine", "layout", "notify", "signal", "mla", "swiglu", "rms", "rope", "ll_", "grouped_gemm", "rowquant",
            "skinny", "bmm_rope", "pre_", "gqa", "gemm2", "selftest", "add_", "split_", "decode_plan")
    rows = sorted(((k, v) for k, v in agg.items() if k[0].startswith(ours)), key=lambda kv: (kv[0][0], -kv[0][1][0] * max(kv[0][1][1], 1)))
    with open(os.path.join(out_dir, f"r{rnd}_kernel_stats_by_grid.csv"), "w") as o:
        o.write("kernel,grid_threads_x,grid_y,grid_z,workgroup_x,calls,avg_ns,p50_ns,min_ns,max_ns\n")
        for (name, grid, wg), v in rows:
            v = sorted(v)
            o.write(f"\"{name}\",{grid[0]},{grid[1]},{grid[2]},{wg[0]},{len(v)},{sum(v) / len(v):.0f},{v[len(v) // 2]},{v[0]},{v[-1]}\n")


def main():
    rnd, stats_dir = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    f = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out_dir, f"r{rnd}_kernel_stats.csv"), "w") as o:
        o.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
        for r in rows[:25]:
            o.write(f"{short(r['Name'])},{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},"
                    f"{r['MinNs']},{r['MaxNs']}\n")
    by_grid(rnd, stats_dir, out_dir)
    if len(sys.argv) >= 5:
        agg = defaultdict(lambda: defaultdict(list))
        order = {}                                  # counter -> launches in dispatch order: (kernel, value)
        for d, cname in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
            seq = []
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == cname:
                    agg[short(r["Kernel_Name"])][cname].append(float(r["Counter_Value"]))
                    seq.append((int(r["Dispatch_Id"]), short(r["Kernel_Name"]), float(r["Counter_Value"])))
            order[cname] = [(k, v) for _, k, v in sorted(seq)]
        out = {}
        for k, v in agg.items():
            if not (k.startswith(("stage", "pull", "combine", "layout", "notify", "mla", "swiglu", "rms", "rope", "ll_", "grouped_gemm",
                                  "rowquant", "skinny", "bmm_rope", "pre_", "decode_plan"))):
                continue
            # the same kernel runs at several problem sizes in one bench (C2-size and decode-size launches): price the LARGEST size
            # only -- launches within 20 % of the kernel's biggest counter value -- which is the one bench.py's roofline quotes
            def top(vals):
                if not vals:
                    return 0.0, 0
                m = max(vals)
                sel = [x for x in vals if x >= 0.8 * m]
                return sum(sel) / len(sel), len(sel)
            (fe, nf), (wr, _) = top(v["FETCH_SIZE"]), top(v["WRITE_SIZE"])
            out[k] = {"fetch_size_kib_raw": fe, "write_size_kib_raw": wr, "launches": nf, "launches_all_sizes": len(v["FETCH_SIZE"]),
                      "hbm_bytes_per_launch": 2 * fe * 1024 + wr * 1024}
        # The MLA merge launch runs behind decode launches of very different kinds in one bench (full-length batches whose two-piece sequences
        # finish inside the decode kernel and leave it nothing; ragged batches it merges for real): its figure FOR THE HEADLINE CALL is the
        # launch that follows a largest-size decode launch, not the largest merge launch.
        head = "mla_decode_wide8s_kernel<true, true>"
        if head in out and "mla_merge_kernel<true>" in out:
            per = {}
            for cname, seq in order.items():
                top = max((v for k, v in seq if k == head), default=0.0)
                vals, armed = [], False
                for k, v in seq:
                    if k == head:
                        armed = v >= 0.8 * top
                    elif k == "mla_merge_kernel<true>" and armed:
                        vals.append(v)
                        armed = False
                per[cname] = sum(vals) / len(vals) if vals else None
            if per.get("FETCH_SIZE") is not None and per.get("WRITE_SIZE") is not None:
                out["mla_merge_kernel<true>"]["hbm_bytes_per_launch_behind_headline_decode"] = 2 * per["FETCH_SIZE"] * 1024 + per["WRITE_SIZE"] * 1024
        # every kernel bench.py looks up must be there: a renamed kernel fails the collection instead of leaving a stale file behind
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_names", os.path.join(os.path.dirname(out_dir), "bench.py"))
        src = open(spec.origin).read()
        names = re.search(r"PMC_KERNEL_NAMES = (\{.*?\})", src, re.S)
        wanted = list(eval(names.group(1)).values()) + ["mla_merge_kernel<true>"]
        if not any(k.startswith("mla_decode_wide") for k in out):
            wanted.append("mla_decode_wide_kernel<true>")
        missing = [k for k in wanted if k not in out]
        if missing:
            print("PMC counters are missing kernels that bench.py prices:", missing, "-- have:", sorted(out), file=sys.stderr)
            sys.exit(2)
        json.dump({"note": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction)", "kernels": out},
                  open(os.path.join(out_dir, f"r{rnd}_pmc_traffic.json"), "w"), indent=1)
    print("wrote", os.listdir(out_dir))


if __name__ == "__main__":
    main()
