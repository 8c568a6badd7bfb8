# Fabric traffic (FETCH_SIZE x 2 + WRITE_SIZE, the guide's HBM-side bytes) of the grouped GEMMs inside fused_deep_moe at BASELINE C5, per launch:
# what moved it between rounds 4 and 5?  A/B of the round-5 changes: the small last row block (MI_GEMM_SMALL_LAST) and GEMM1 reading the staged
# rows in place (MI_EP_FUSED_GATHER).  -> gpurun_out/gemm_traffic_ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for cfg in "1 1" "0 1" "1 0"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_gemm_tr/$1$2/$c
    MI_GEMM_SMALL_LAST=$1 MI_EP_FUSED_GATHER=$2 PMC_ONLY_FUSED=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm_tr/$1$2/$c -- python $R/tools/probes/pmc_workload.py > /dev/null 2>&1
  done
done
cd $R
python - <<'PY' | tee gpurun_out/gemm_traffic_ab.txt
import csv, glob, collections
for cfg in ("11", "01", "10"):
    tot = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f'gpurun_out/pmc_gemm_tr/{cfg}/{c}/*/*counter_collection.csv'):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if ('grouped_gemm' in r['Kernel_Name'] or 'rowquant' in r['Kernel_Name'] or 'pull' in r['Kernel_Name']) and r['Counter_Name'] == c:
                    agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
            for k, vals in agg.items():
                vals.sort(); tot[k][c] = vals[len(vals) // 2]
    for k, d in sorted(tot.items()):
        print(f"small_last={cfg[0]} in_place_rows={cfg[1]}", k, {c: round(x / 1e6, 3) for c, x in d.items()},
              "-> 2*FETCH+WRITE =", round((2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / 1e9, 3), "GB")
PY
