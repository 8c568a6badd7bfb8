# fused_deep_moe at BASELINE C5 with the staged rows multiplied in place vs the K-fold gathered copy, alternating on one box
for rep in 1 2; do for m in 0 1; do export MI_EP_FUSED_GATHER=$m; python bench.py 2>/dev/null | tail -1 | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); f=d['fused_deep_moe']; k=f['kernels_avg_us']
print('gather  ' if os.environ['MI_EP_FUSED_GATHER']=='0' else 'in place', 'C5 ms', round(f['ms_p50'],4), 'frac', round(f['roofline']['frac'],4), 'step', round(d['ms_per_step'],4), {n:round(v,1) for n,v in k.items() if 'gemm' in n or 'pull' in n or 'resolve' in n or 'stage' in n})
"; done; done
