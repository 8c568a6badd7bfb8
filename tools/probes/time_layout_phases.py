"""Scratch: phase timestamps (100 MHz) of the cooperative layout kernel (library built with -DLAYOUT_TIMING)."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, stream_ptr
from ctypes import c_int, c_size_t, c_void_p
L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "timing_ep", "libmi_ep_t.so"))
V, I = c_void_p, c_int
L.mi_ep_dispatch_layout.argtypes = [V, I, I, I, I, I, V, V, V, V, V, V, c_size_t, V, V]
for T in (128, 1024, 4096):
    E, W, K = 256, 8, 8
    idx = torch.topk(torch.rand((T, E), device="cuda"), K, dim=-1)[1]
    i32 = dict(dtype=torch.int32, device="cuda")
    o = [torch.empty(W, **i32), torch.empty(E, **i32), torch.empty((T, W), **i32), torch.empty((T, K), **i32), torch.empty(E, **i32)]
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    words = torch.zeros(2, **i32)
    for _ in range(5):
        L.mi_ep_dispatch_layout(ptr(idx), 0, T, K, E, W, *[ptr(t) for t in o], ptr(ws), 1 << 20, ptr(words), stream_ptr())
    torch.cuda.synchronize()
    tk = ws[512 << 10:(512 << 10) + 64].view(torch.int64).cpu().tolist()
    d = [(tk[i + 1] - tk[i]) / 100 for i in range(5)]
    print(f"T={T}: us [init+pass1, totals+grid barrier, pass2, pass3+tail] =", [round(v, 2) for v in d[:4]], "total", round(sum(d[:4]), 2), flush=True)
