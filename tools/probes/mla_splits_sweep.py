"""Scratch: C4 MLA decode (full and ragged) against the number of KV splits."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
import sgl_kernel_npu
for ragged in (False, True):
    q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64, ragged=ragged)
    out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device="cuda")
    res = {}
    for splits in (0, 1, 2, 3, 4, 6, 8):
        f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, splits)
        for _ in range(60): f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): f()
        b.record(); torch.cuda.synchronize()
        res[splits] = round(a.elapsed_time(b) / 100 * 1e3, 1)
    print("ragged" if ragged else "full  ", res, flush=True)
