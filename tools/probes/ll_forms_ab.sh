# scratch: low-latency section of bench.py under the launch forms (MI_EP_LL_FUSED_COUNTS x MI_EP_COMBINE_FUSED), same box, alternating
cd $GRAFT_REPO_ROOT
P='import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l)["low_latency"]; g=d["graph_replay"]; print(sys.argv[1], "dispatch p50 %.1f combine p50 %.1f | queued %.1f %.1f | graph %.1f %.1f pair-of-10 %.1f ok %s" % (d["dispatch_us_p50"], d["combine_us_p50"], d["queued"]["dispatch_us_p50"], d["queued"]["combine_us_p50"], g["dispatch_us_p50"], g["combine_us_p50"], g["pair_us_p50_in_graph_of_10"], d["validated_round_trip"]))'
for rep in 1 2; do
  for v in ${FORMS:-"0 0" "2 2"}; do
    set -- $v
    MI_EP_LL_FUSED_COUNTS=$1 MI_EP_COMBINE_FUSED=$2 python bench.py --no-mla --no-cpu-baseline 2>/dev/null | python -c "$P" dispatch_$1_combine_$2
  done
done
