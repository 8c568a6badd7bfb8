"""Scratch: idle gaps between consecutive kernels of the normal-mode step, from a rocprofv3 --kernel-trace CSV.
   python tools/probes/gap_analysis.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].split("::")[-1][:40]
# steady state: the last 25 occurrences of combine_reduce delimit the steps
idx = [i for i, r in enumerate(rows) if "combine_reduce_kernel" in r["Kernel_Name"]]
# bench.py order: first call, warm-up, K un-instrumented steps, then the same again with HIP events around every kernel (which
# opens ~10 us gaps): argv[2] picks the window of reduce launches to analyse (default: launches 10 .. 35 = inside the timed pass)
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10, 36)
idx = idx[lo:hi]
gaps, durs = defaultdict(list), defaultdict(list)
for a, b in zip(idx[:-1], idx[1:]):
    for i in range(a + 1, b + 1):
        prev, cur = rows[i - 1], rows[i]
        gaps[short(cur["Kernel_Name"])].append((int(cur["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3)
        durs[short(cur["Kernel_Name"])].append((int(cur["End_Timestamp"]) - int(cur["Start_Timestamp"])) / 1e3)
step = (int(rows[idx[-1]]["End_Timestamp"]) - int(rows[idx[0]]["End_Timestamp"])) / 1e3 / (len(idx) - 1)
print(f"step {step:.1f} us")
tg = td = 0
for k in gaps:
    g, d = sum(gaps[k]) / (len(idx) - 1), sum(durs[k]) / (len(idx) - 1)
    tg += g; td += d
    print(f"{k:42s} n/step {len(gaps[k]) / (len(idx) - 1):.1f}  dur {d:7.1f} us  gap before {g:6.1f} us")
print(f"sum dur {td:.1f}  sum gap {tg:.1f}")
