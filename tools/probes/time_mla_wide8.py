"""Scratch: per-phase clock breakdown of the eight-wave wide MLA kernel (library built with -DMLA8_TIMING: pass its path)."""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
from capi import ptr, stream_ptr
B, Hq, S, page = 128, 128, 4096, 64
q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page, ragged=len(sys.argv) > 3)
out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
L = ctypes.CDLL(sys.argv[1])
L.mi_mla_decode_workspace.restype = c_size_t
L.mi_mla_decode.argtypes = [c_void_p] * 6 + [c_int] * 6 + [c_int64] * 10 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]
splits = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wsb = L.mi_mla_decode_workspace(B, Hq, splits)
ws = torch.zeros(wsb + (1 << 20), dtype=torch.uint8, device="cuda")
for _ in range(300):
    L.mi_mla_decode(ptr(q), ptr(kn), ptr(kr), ptr(out), ptr(lens), ptr(bt), B, Hq, 1, page, bt.stride(0), S, q.stride(0), q.stride(1),
                    kn.stride(0), kn.stride(1), kn.stride(2), kr.stride(0), kr.stride(1), kr.stride(2), out.stride(0), out.stride(1),
                    576 ** -0.5, 0, splits, ptr(ws), wsb, stream_ptr())
torch.cuda.synchronize()
part = B * Hq * splits * 514 * 4 if splits > 1 else 0
dbg = ws[part + 2048 * 4: part + 2048 * 4 + 64 * 8 * 8 * 4].view(torch.float32).reshape(64, 8, 8).cpu()
m = dbg.mean(dim=(0, 1)).tolist()
print("per tile [own vmcnt wait, barrier A + addresses, QK^T, softmax + publish, barrier B, P.V] (s_memtime ticks):", [round(v) for v in m[:6]], "sum", round(sum(m[:6])))
print("entry -> loop end: s_memtime ticks", round(m[6]), " s_memrealtime ticks", round(m[7]), " ratio", m[6] / max(m[7], 1))
for w in range(8):
    print("wave", w, [round(v) for v in dbg[:, w, :6].mean(dim=0).tolist()])
# workgroup-level stamps (100 MHz): [start, Q^T resident, loop end, exit] per workgroup, relative to the earliest start
nwg = B * splits
off = part + 2048 * 4 + 64 * 8 * 8 * 4
st = ws[off: off + nwg * 32].view(torch.int64).reshape(nwg, 4).cpu().double()
t0 = st[:, 0].min()
st = (st - t0) / 100.0       # us
import numpy as np
a = st.numpy()
pct = lambda v: [round(float(np.percentile(v, q)), 2) for q in (0, 50, 90, 100)]
print("us since first workgroup start, percentiles [min, p50, p90, max]:")
print("  start        ", pct(a[:, 0]))
print("  Q^T resident ", pct(a[:, 1]), " prologue per WG", pct(a[:, 1] - a[:, 0]))
print("  loop end     ", pct(a[:, 2]), " loop per WG    ", pct(a[:, 2] - a[:, 1]))
print("  exit         ", pct(a[:, 3]), " epilogue per WG", pct(a[:, 3] - a[:, 2]))
