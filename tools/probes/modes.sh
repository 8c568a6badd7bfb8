# scratch: how often does a fresh process land in the slow mode (push ~172 us) vs the fast one (~158 us)?
P='import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l); print(round(d["ms_per_step"],4), {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})'
for i in $(seq 1 ${1:-8}); do
  python bench.py --steps 60 --warmup 60 --no-mla --no-cpu-baseline 2>/dev/null | python -c "$P"
done
