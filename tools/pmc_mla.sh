cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_mla/$n -- python /root/repo/tools/bench_mla_variants.py libmi_sgl_kernels.so > /dev/null 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_mla/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'wide' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        v.sort(); print(k, 'median', v[len(v)//2], 'n', len(v))
PY
