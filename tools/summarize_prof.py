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
    if len(sys.argv) >= 5:
        agg = defaultdict(lambda: defaultdict(list))
        for d, cname in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == cname:
                    agg[short(r["Kernel_Name"])][cname].append(float(r["Counter_Value"]))
        out = {}
        for k, v in agg.items():
            if not (k.startswith(("stage", "pull", "combine", "layout", "notify", "mla", "swiglu", "rms", "rope", "ll_"))):
                continue
            fe = sum(v["FETCH_SIZE"]) / max(len(v["FETCH_SIZE"]), 1)
            wr = sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
            out[k] = {"fetch_size_kib_raw": fe, "write_size_kib_raw": wr, "launches": len(v["FETCH_SIZE"]),
                      "hbm_bytes_per_launch": 2 * fe * 1024 + wr * 1024}
        json.dump({"note": "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction)", "kernels": out},
                  open(os.path.join(out_dir, f"r{rnd}_pmc_traffic.json"), "w"), indent=1)
    print("wrote", os.listdir(out_dir))


if __name__ == "__main__":
    main()
