"""Scratch: A/B the MLA kernel build variants (lib/libmla_w{4,8}.so) through the C-ABI."""
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
ref = None
for name in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", name))
    L.mi_mla_decode_workspace.restype = c_size_t
    L.mi_mla_decode.argtypes = [c_void_p] * 6 + [c_int] * 6 + [c_int64] * 10 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]
    for splits in (1, 2):
        wsb = L.mi_mla_decode_workspace(B, Hq, splits)
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
        f = lambda: L.mi_mla_decode(ptr(q), ptr(kn), ptr(kr), ptr(out), ptr(lens), ptr(bt), B, Hq, 1, page, bt.stride(0), S,
                                    q.stride(0), q.stride(1), kn.stride(0), kn.stride(1), kn.stride(2), kr.stride(0), kr.stride(1),
                                    kr.stride(2), out.stride(0), out.stride(1), 576 ** -0.5, 0, splits, ptr(ws), wsb, stream_ptr())
        for _ in range(30): assert f() == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 50 * 1e3
        if ref is None: ref = out.clone()
        err = (out.float() - ref.float()).abs().max().item()
        print(f"{name} splits={splits}: {us:.1f} us  ({(B*S*1152 + B*Hq*2176)/us/1e3:.0f} GB/s, {B*Hq*S*1088*2/us/1e6:.0f} TFLOP/s) maxdiff_vs_first={err:.2e}", flush=True)
