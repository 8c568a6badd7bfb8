# scratch: L1<-L2 request counters of the grouped GEMM (is the 64-byte k-slice fetched as half a 128-byte line twice?)
cd /tmp && export TMPDIR=/tmp
for c in "TCP_TCC_READ_REQ_sum" "TCC_READ_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_gemm/$n -- python /root/repo/tools/time_gemm.py libmi_ep.so > /dev/null 2>&1
done
cd /root/repo
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_gemm/*/*/*counter_collection.csv')):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'grouped_gemm' in r['Kernel_Name']:
            agg[(r['Counter_Name'], r['Kernel_Name'][:60])].append(float(r['Counter_Value']))
    for k, v in agg.items():
        v.sort(); print(k, 'median', v[len(v)//2], 'n', len(v))
PY
