# scratch: combine_reduce time vs waves per token (MI_EP_REDUCE_WAVES = target wave count; segments per token double until it is met)
P='import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],4), {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})'
for w in 2048 8192 16384 32768 65536; do
  MI_EP_REDUCE_WAVES=$w python bench.py --steps 60 --warmup 10 --no-mla --no-cpu-baseline 2>/dev/null | python -c "$P" $w
done
