#!/bin/bash
# usage: bash tools/probes/prof_any.sh <python script> [args]   -> per-kernel stats (rocprofv3 --kernel-trace --stats), top rows
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_any
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_any -- python "$GRAFT_REPO_ROOT/$1" "${@:2}" > /dev/null 2>&1
f=$(find /tmp/prof_any -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    n = re.sub(r"\(.*", "", r["Name"])[-70:]
    print(f"{n:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}  max {float(r['MaxNs'])/1e3:8.1f}")
PY
t=$(find /tmp/prof_any -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, re
from collections import defaultdict
agg = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"])[-50:]
    if "mi_" not in r["Kernel_Name"]: continue
    agg[(n, int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (n, g, w), v in sorted(agg.items()):
    v.sort()
    print(f"{n:50s} grid {g:8d} wg {w:5d} calls {len(v):5d} p50 {v[len(v)//2]/1e3:8.1f} us min {v[0]/1e3:8.1f}")
PY
