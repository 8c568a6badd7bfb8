"""Scratch: time the MLA decode op over head counts / batch sizes (device events around torch.ops.npu.decode_mla, auto splits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu
from sgl_kernel_npu.bench_hooks import _mla_inputs

def main():
    page = 64
    for B, Hq, S in ((128, 16, 4096), (128, 32, 4096), (128, 64, 4096), (128, 128, 4096), (32, 128, 4096), (16, 128, 8192), (256, 128, 2048)):
        q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
        out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
        f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, page, bt, 0)
        for _ in range(30): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 30 * 1e3
        print(f"B={B} Hq={Hq} S={S}: {us:.1f} us  ({(B*S*1152 + B*Hq*2176)/us/1e3:.0f} GB/s, {B*Hq*S*1088*2/us/1e6:.0f} TFLOP/s)", flush=True)
        del q, kn, kr, bt, lens, out

main()
