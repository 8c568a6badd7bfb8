# Scratch: write-back share of the receive buffer at other batch sizes (default 80 MiB share, all write-back, all around the cache)
for t in 2048 8192; do for nt in d 0 1; do
  echo "T=$t NT=$nt"
  if [ $nt = d ]; then unset MI_EP_PULL_NT; else export MI_EP_PULL_NT=$nt; fi
  python bench.py --tokens $t --no-mla --no-extra --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done; done
