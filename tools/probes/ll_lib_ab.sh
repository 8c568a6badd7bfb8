# scratch: low-latency section of bench.py with the product libmi_ep.so vs LD_PRELOADed builds of it (tools/build_timing_ep.sh <sfx>), alternating
# usage: SFX="oldsend ntload" bash tools/probes/ll_lib_ab.sh
cd $GRAFT_REPO_ROOT
P='import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l)["low_latency"]; g=d["graph_replay"]; print(sys.argv[1], "dispatch p50 %.1f combine p50 %.1f | queued %.1f %.1f | graph %.1f %.1f pair-of-10 %.1f ok %s" % (d["dispatch_us_p50"], d["combine_us_p50"], d["queued"]["dispatch_us_p50"], d["queued"]["combine_us_p50"], g["dispatch_us_p50"], g["combine_us_p50"], g["pair_us_p50_in_graph_of_10"], d["validated_round_trip"]))'
for rep in 1 2 3; do
  python bench.py --no-mla --no-cpu-baseline 2>/dev/null | python -c "$P" product
  for s in $SFX; do
    LD_PRELOAD=sgl-kernel-npu_amd/lib/timing_ep/libmi_ep_$s.so python bench.py --no-mla --no-cpu-baseline 2>/dev/null | python -c "$P" $s
  done
done
