# scratch: A/B libmi_ep build variants through bench.py (variant libs are lib/libmi_ep_<name>.so)
P='import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l); print(sys.argv[1], round(d["ms_per_step"],4), d["validated_round_trip"], {k:round(v["avg_us"],1) for k,v in d["kernels"].items()})'
cp sgl-kernel-npu_amd/lib/libmi_ep.so /tmp/libmi_ep_orig.so
for v in "$@"; do
  cp sgl-kernel-npu_amd/lib/libmi_ep_$v.so sgl-kernel-npu_amd/lib/libmi_ep.so
  python bench.py --steps 100 --warmup 100 2>/dev/null | python -c "$P" $v
done
cp /tmp/libmi_ep_orig.so sgl-kernel-npu_amd/lib/libmi_ep.so
