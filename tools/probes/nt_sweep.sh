for t in 1024 2048; do for nt in 0 1; do
  echo "T=$t NT=$nt"
  MI_EP_PULL_NT=$nt python bench.py --tokens $t --no-mla --no-extra --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done
