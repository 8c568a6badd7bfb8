# scratch: grouped GEMM at C5 shapes with and without XCD-consistent column tiles (GEMM2: 28 column tiles): time, shader clock, fabric traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
for v in 0 1; do
  echo "MI_GEMM_XCD_COLS=$v"
  MI_GEMM_XCD_COLS=$v python $R/tools/time_gemm.py libmi_ep.so 2>&1 | grep -v amdgpu.ids
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_gemm_ab/$v/$c
    MI_GEMM_XCD_COLS=$v rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm_ab/$v/$c -- python $R/tools/time_gemm.py libmi_ep.so > /dev/null 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
for v in (0, 1):
    tot = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f'gpurun_out/pmc_gemm_ab/{v}/{c}/*/*counter_collection.csv'):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if 'grouped_gemm' in r['Kernel_Name'] and r['Counter_Name'] == c:
                    agg[r['Kernel_Name'][:48]].append(float(r['Counter_Value']))
            for k, vals in agg.items():
                vals.sort(); tot[k][c] = vals[len(vals) // 2]
    for k, d in tot.items():
        print("xcd_cols", v, k, {c: round(x / 1e6, 3) for c, x in d.items()}, "GB: hbm-side bytes = 2*FETCH+WRITE KiB ->",
              round((2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / 1e9, 3), "GB")
PY
