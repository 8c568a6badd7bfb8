"""A/B at BASELINE C4 (bs 128 x 128 heads x 4096 keys): the wide kernel (128 heads per workgroup, 2 splits + merge) against the 64-head
kernel run as two XCD-paired workgroups per sequence (MI_MLA_WIDE=0), with 1 and 2 splits.  One process per variant (the switch is
read once)."""
import os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, "..", "..", "sgl-kernel-npu_amd", "python"))
from sgl_kernel_npu.bench_hooks import _mla_inputs
B, Hq, S, page = 128, 128, 4096, 64
q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
sm = 576 ** -0.5
splits = int(sys.argv[1])
f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, sm, page, bt, splits)
import sgl_kernel_npu
for _ in range(300): f()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 50 * 1e3)
byts = float(lens.sum().item()) * 576 * 2 + B * Hq * (576 + 512) * 2
us = sorted(ts)[len(ts) // 2]
print(f"MI_MLA_WIDE={os.getenv('MI_MLA_WIDE', '1')} splits={splits}: {us:.1f} us  {byts / us / 1e6:.2f} TB/s  frac {byts / us / 1e6 / 8:.3f}", flush=True)
''' % HERE
variants = [("1", "1", 0), ("1", "0", 0), ("1", "1", 1), ("1", "1", 3), ("1", "1", 4)] if "--wide8" in sys.argv else \
    [("1", "1", 0), ("0", "1", 1), ("0", "1", 2), ("1", "0", 1), ("0", "1", 4)]
for wide, wide8, splits in variants:
    env = dict(os.environ, MI_MLA_WIDE=wide, MI_MLA_WIDE8=wide8)
    print(f"MI_MLA_WIDE8={wide8}", end=" ", flush=True)
    subprocess.run([sys.executable, "-c", CHILD, str(splits)], env=env)
