#!/bin/bash
# usage: bash tools/probes/mla_fetch.sh <variant>  -> HBM / L2 counters of the MLA decode kernel at C4
cd /tmp && export TMPDIR=/tmp
for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_BUBBLE_sum"; do
  rm -rf /tmp/pf; rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pf -- python "$GRAFT_REPO_ROOT/tools/probes/mla_variant_loop.py" $1 12 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pf/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "mla_decode" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v.sort(); print(k, "median", v[len(v)//2], "n", len(v))
PY
done
