"""Scratch: per-phase shader-clock breakdown of the wide MLA kernel (library built with -DMLAW_TIMING)."""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
from capi import ptr, stream_ptr
B, Hq, S, page = 128, 128, 4096, 64
q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", sys.argv[1]))
L.mi_mla_decode_workspace.restype = c_size_t
L.mi_mla_decode.argtypes = [c_void_p] * 6 + [c_int] * 6 + [c_int64] * 10 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]
splits = 2
wsb = L.mi_mla_decode_workspace(B, Hq, splits)
ws = torch.zeros(wsb + (1 << 20), dtype=torch.uint8, device="cuda")
for _ in range(3):
    L.mi_mla_decode(ptr(q), ptr(kn), ptr(kr), ptr(out), ptr(lens), ptr(bt), B, Hq, 1, page, bt.stride(0), S, q.stride(0), q.stride(1),
                    kn.stride(0), kn.stride(1), kn.stride(2), kr.stride(0), kr.stride(1), kr.stride(2), out.stride(0), out.stride(1),
                    576 ** -0.5, 0, splits, ptr(ws), wsb, stream_ptr())
torch.cuda.synchronize()
part = B * Hq * splits * 514 * 4
dbg = ws[part + 4096: part + 4096 + 64 * 4 * 4 * 4].view(torch.float32).reshape(64, 4, 4).cpu()
print("per-tile s_memtime ticks (100 MHz => x10 ns) [top-wait, qk, softmax+pv]:")
print("mean over 64 WGs x 4 waves:", dbg[:, :, :3].mean(dim=(0, 1)).tolist())
print("wave0..3 of WG0:", dbg[0, :, :3].tolist())
dbg2 = ws[part + 4096 + 4096: part + 4096 + 4096 + 64 * 4 * 4 * 4].view(torch.float32).reshape(64, 4, 4).cpu()
print("whole kernel, cycles [entry -> end of tile loop, epilogue incl. drain]:", dbg2[:, :, :2].mean(dim=(0, 1)).tolist())
