"""Scratch: four-wave vs eight-wave (8: block-id ring, 9: scalar block ids) wide MLA kernels at BASELINE C4, alternating in ONE process
(same box, same clocks); num_splits 0 = each form's default (planned list for the eight-wave forms, two uniform splits for four waves)."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
import sgl_kernel_npu
lib = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_sgl_kernels.so"), mode=ctypes.RTLD_GLOBAL)
for ragged in (False, True):
    q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64, ragged=ragged)
    out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device="cuda")
    f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, 0)
    for _ in range(300): f()
    res = {4: [], 8: [], 9: []}
    for rep in range(6):
        for waves in (4, 8, 9):
            lib.mi_mla_decode_select_wide(waves)
            for _ in range(20): f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100): f()
            b.record(); torch.cuda.synchronize()
            res[waves].append(a.elapsed_time(b) / 100 * 1e3)
    print("ragged" if ragged else "full  ", {k: [round(x, 1) for x in v] for k, v in res.items()}, flush=True)
