# scratch: fabric-side read traffic and L2 hit rate of the mla_preprocess kernels (do the two row halves of a head share an L2?)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/pmc_mla_pre
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MLA_PRE_QUANT=per_tensor_quant_asymm
i=0
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$REPO/tools/probes/time_mla_pre_op.py" > /dev/null 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if any(k in n for k in ("gemm2_bmm", "pre_mid", "skinny", "pre_quant", "one_launch")):
            agg[(re.sub(r"\(.*", "", n)[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v.sort(); print(k, "median %.4g" % v[len(v) // 2], "n", len(v))
PY
