"""Scratch: phase stamps of decode_plan_kernel (library built with -DPLAN_TIMING, LD_PRELOADed): C4 uniform and ragged lengths."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, stream_ptr
from sgl_kernel_npu.bench_hooks import _mla_inputs
L = ctypes.CDLL(os.environ["LD_PRELOAD"].split(":")[0])
L.mi_mla_decode_plan_bytes.restype = ctypes.c_size_t
for ragged in (False, True):
    _, _, _, _, lens = _mla_inputs(128, 128, 4096, 64, ragged=ragged)
    nb = L.mi_mla_decode_plan_bytes(128, 1)
    plan = torch.zeros(nb // 4, dtype=torch.int32, device="cuda")
    for _ in range(20):
        assert L.mi_mla_decode_build_plan(ptr(lens), 128, 1, ptr(plan), ctypes.c_size_t(nb), stream_ptr()) == 0
    torch.cuda.synchronize()
    t = plan[8:13].cpu().tolist()
    print("ragged" if ragged else "uniform", "us between stamps [entry->totals, ->piece size, ->ranks, ->bases, ->end]:", [round((b - a) / 100.0, 2) for a, b in zip(t, t[1:])], "total", (t[4] - t[0]) / 100.0)
