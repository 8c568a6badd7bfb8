"""Scratch: per-k-tile wait / compute cycles of the grouped GEMM (libmi_ep built with -DGEMM_TIMING)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, stream_ptr
L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", sys.argv[1]))
E, H, I2, M = 32, 7168, 4096, 32768
a = torch.randint(-8, 8, (M, H), dtype=torch.int8, device="cuda")
asc = torch.rand(M, device="cuda")
w = torch.randint(-8, 8, (E, I2, H), dtype=torch.int8, device="cuda")
ws = torch.rand((E, I2), device="cuda")
cum = (torch.arange(1, E + 1, device="cuda", dtype=torch.int32) * (M // E)).contiguous()
out = torch.zeros((M, I2 // 2), dtype=torch.float32, device="cuda")
c_vp = ctypes.c_void_p
L.mi_ep_moe_gemm1_swiglu.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, c_vp]
f = lambda: L.mi_ep_moe_gemm1_swiglu(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, H, I2, ptr(out), 0, stream_ptr())
for _ in range(3): assert f() == 0
torch.cuda.synchronize()
for rep in range(8):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 40 * 1e3
    print(f"rep {rep}: gemm1 {us:.1f} us  {2 * M * H * I2 / us / 1e6:.0f} TOPS", flush=True)
import numpy as np
buf = np.zeros(128, dtype=np.float32)
L.mi_ep_gemm_dbg(buf.ctypes.data_as(c_vp))
dbg = torch.from_numpy(buf).reshape(-1, 2)[:32]
print("per k-tile cycles [wait, compute] mean:", dbg.mean(dim=0).tolist())
