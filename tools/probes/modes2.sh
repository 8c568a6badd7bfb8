# scratch: does the step time depend on how long the process has been running (clock / power state)?
P='import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],4), {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})'
for w in 5 100 500 2000 5 2000; do
  python bench.py --steps 100 --warmup $w --no-mla --no-cpu-baseline 2>/dev/null | python -c "$P" $w
done
