# fused_deep_moe at BASELINE C5 with the requantisation in GEMM1's epilogue (MI_EP_FUSED_REQUANT=1) vs the rowquant launch (=0), alternating on one box
for rep in 1 2; do for m in 0 1; do export MI_EP_FUSED_REQUANT=$m; python bench.py 2>/dev/null | tail -1 | python -c "
import sys,json,os
d=json.loads(sys.stdin.read()); f=d['fused_deep_moe']; k=f['kernels_avg_us']
print('rowquant launch' if os.environ['MI_EP_FUSED_REQUANT']=='0' else 'in the epilogue', 'C5 ms', round(f['ms_p50'],4), 'frac', round(f['roofline']['frac'],4), 'step', round(d['ms_per_step'],4), {n:round(v,1) for n,v in k.items() if 'gemm' in n or 'quant' in n})
"; done; done
