"""Scratch: time the MLA decode kernel at BASELINE C4 for several split counts (device events around the op)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu
from sgl_kernel_npu.bench_hooks import _mla_inputs

def main():
    B, Hq, S, page = 128, 128, 4096, 64
    q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    for splits in (1, 2, 4, 8):
        f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, page, bt, splits)
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        print(f"splits={splits}: {us:.1f} us  ({(B*S*1152 + B*Hq*2176)/us/1e3:.0f} GB/s, {B*Hq*S*1088*2/us/1e6:.0f} TFLOP/s)", flush=True)

main()
