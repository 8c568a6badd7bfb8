#!/bin/bash
# per-kernel split of mla_preprocess (128 tokens x 128 heads): rocprofv3 kernel stats of tools/probes/time_mla_pre.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $R/tools/probes/time_mla_pre.py > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("/tmp/pp/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 40: print(r["Name"][:90], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
