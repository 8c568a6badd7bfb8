"""Scratch: MLA decode time vs KV split count for small batches (is the heuristic in mi_mla_decode_num_splits right?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu
from sgl_kernel_npu.bench_hooks import _mla_inputs
page = 64
for B, Hq, S in ((32, 128, 4096), (16, 128, 8192), (64, 128, 4096), (8, 128, 16384), (32, 16, 4096)):
    q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    res = []
    for splits in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, page, bt, splits)
        for _ in range(20): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): f()
        b.record(); torch.cuda.synchronize()
        res.append((splits, round(a.elapsed_time(b) / 30 * 1e3, 1)))
    print(f"B={B} Hq={Hq} S={S}:", res, flush=True)
    del q, kn, kr, bt, lens, out
