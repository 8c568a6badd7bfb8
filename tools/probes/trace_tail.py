"""Scratch: average duration / preceding gap per kernel over the last N kernels of a rocprofv3 kernel trace CSV."""
import csv, sys
from collections import OrderedDict
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n:]
agg = OrderedDict()
for p, c in zip(rows[:-1], rows[1:]):
    k = c["Kernel_Name"].split("(")[0].split("::")[-1][:44]
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (int(c["End_Timestamp"]) - int(c["Start_Timestamp"])) / 1e3; a[2] += (int(c["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3
for k, (c, d, g) in agg.items():
    print(f"{k:46s} n {c:4d}  dur {d / c:7.1f} us  gap before {g / c:6.1f} us")
print("span per kernel-sequence repeat:", (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3, "us over", len(rows), "kernels")
