# scratch: instruction-fetch counters of the mla_preprocess kernels (straight-line code executed once per workgroup)
REPO=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$REPO/gpurun_out/pmc_mla_pre_if
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MLA_PRE_QUANT=per_tensor_quant_asymm
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|IFETCH|inst_cache|SQ_WAIT|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_INSTS_VALU |SQ_INST_LEVEL" | cut -c1-160 | sort -u | head -40
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVES" "SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READYB"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$REPO/tools/probes/time_mla_pre_op.py" > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
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
