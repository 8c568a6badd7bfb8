# sweep the workgroup caps of the two big copy kernels (rows are dealt round-robin over the grid)
for v in 512 1024 2048 4096 8192; do
  MI_EP_PUSH_BLOCKS=$v MI_EP_PULL_BLOCKS=$v python bench.py --no-mla --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print($v, round(d['ms_per_step'],4), 'pull', round(d['kernels']['dispatch_pull']['avg_us'],1), 'push', round(d['kernels']['combine_push']['avg_us'],1), 'reduce', round(d['kernels']['combine_reduce']['avg_us'],1))"
done
