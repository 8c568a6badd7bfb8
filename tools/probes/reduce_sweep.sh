for v in 2048 8192 16384 32768 65536; do
  MI_EP_REDUCE_WAVES=$v python bench.py --no-mla --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],4), round(d['kernels']['combine_reduce']['avg_us'],1), round(d['kernels']['combine_push']['avg_us'],1))"
done
