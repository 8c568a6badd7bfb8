"""Scratch: 100 MHz stamps of the eight-wave scalar-id MLA kernel's epilogue (library built with tools/build_timing.sh <sfx> -DMLA8S_STAMPS:
pass its path).  BASELINE C4, planned form; stamps per workgroup: 0 start, 1 loop end, 2 partial stores issued, 3 drained + barrier,
4 partner's word seen, 5 acquire done, 6 second pass done."""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
from capi import ptr, stream_ptr
B, Hq, S, page = 128, 128, 4096, 64
L = ctypes.CDLL(sys.argv[1])
L.mi_mla_decode_workspace.restype = c_size_t
L.mi_mla_decode.argtypes = [c_void_p] * 6 + [c_int] * 6 + [c_int64] * 10 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]
L.mi_mla_decode_set_pair.argtypes = [c_int]
for ragged in (False, True):
    q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page, ragged=ragged)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    wsb = L.mi_mla_decode_workspace(B, Hq, -1)
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    for mode in (1, 0):
        L.mi_mla_decode_set_pair(mode)
        def call():
            rc = L.mi_mla_decode(ptr(q), ptr(kn), ptr(kr), ptr(out), ptr(lens), ptr(bt), B, Hq, 1, page, bt.stride(0), S, q.stride(0), q.stride(1),
                                 kn.stride(0), kn.stride(1), kn.stride(2), kr.stride(0), kr.stride(1), kr.stride(2), out.stride(0), out.stride(1),
                                 576 ** -0.5, 0, -1, ptr(ws), wsb, stream_ptr())
            assert rc == 0
        for _ in range(300):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            call()
        b.record()
        torch.cuda.synchronize()
        host = np.zeros((1024, 8), dtype=np.uint64)
        assert L.mi_mla8s_stamps(host.ctypes.data_as(c_void_p)) == 0
        st = host[:256].astype(np.float64)
        t0 = st[:, 0].min()
        st = (st - t0) / 100.0
        pct = lambda v: [round(float(np.percentile(v, q)), 2) for q in (0, 50, 90, 100)]
        print("ragged" if ragged else "full", "pair" if mode else "merge", "us/call", round(a.elapsed_time(b) * 10, 1))
        ph = np.zeros((256, 8, 8), dtype=np.float32)
        assert L.mi_mla8s_phases(ph.ctypes.data_as(c_void_p)) == 0
        m = ph.mean(axis=(0, 1))
        if m[7] == 0:
            m[7] = 1
        print("   per tile and wave [own fill wait, barrier A, QK^T, softmax + publish, barrier B, P.V] shader clocks:", [int(v) for v in m[:6]], "sum", int(m[:6].sum()),
              " loop", int(m[6]), "clocks /", int(m[7]), "tiles")
        for w in range(8):
            print("      wave", w, [int(v) for v in ph[:, w, :6].mean(axis=0)])
        ghz = host[:256, 7].astype(np.float64) / np.maximum((host[:256, 1].astype(np.float64) - host[:256, 0].astype(np.float64)) * 10.0, 1.0)
        print("   shader clock, start -> loop end (GHz) [min, p50, max]:", [round(float(np.percentile(ghz, q)), 3) for q in (0, 50, 100)])
        names = ["start", "loop end", "stores issued", "drained+barrier", "flag seen", "acquired", "second pass done"]
        for i in range(1, 7 if mode else 3):
            print(f"   {names[i]:18s} at {pct(st[:, i])}   step {pct(st[:, i] - st[:, i - 1])}")
