# Scratch: first MB of the receive buffer written back, the rest written around the cache (pull_local, C2 shapes)
for mb in ${MBS:-0 64 112 160}; do
  echo "NT_FROM_MB=$mb"
  MI_EP_PULL_NT_FROM_MB=$mb python bench.py --no-mla --no-extra --no-cpu-baseline --steps 60 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"
done
