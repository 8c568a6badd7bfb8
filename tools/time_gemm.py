"""Scratch: grouped INT8 GEMM timing at the C5 shapes (and, in a -DGEMM_TIMING build, per-k-tile wait / compute cycles)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, stream_ptr
L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", sys.argv[1]))
E, H, I2, M = 32, 7168, 4096, 32768
c_vp = ctypes.c_void_p
sig = [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_int, c_vp]
L.mi_ep_moe_gemm1_swiglu.argtypes = sig
L.mi_ep_moe_gemm2.argtypes = sig
asc = torch.rand(M, device="cuda")
cum_u = (torch.arange(1, E + 1, device="cuda", dtype=torch.int32) * (M // E)).contiguous()
# what routing produces: top-8 of 256 experts over 8 ranks -> multinomial row counts (every expert ends in a partial 256-row tile)
gen = torch.Generator().manual_seed(3)
cnt = torch.bincount(torch.multinomial(torch.ones(E), M, replacement=True, generator=gen), minlength=E)
cum_r = torch.cumsum(cnt, 0).to(torch.int32).cuda().contiguous()
has_q = hasattr(L, "mi_ep_moe_gemm1_swiglu_quant")
if has_q:
    L.mi_ep_moe_gemm1_swiglu_quant.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp] + [ctypes.c_int] * 5 + [c_vp, c_vp, c_vp, ctypes.c_int, c_vp,
                                                                                                  ctypes.c_int, ctypes.c_int, c_vp]
    L.mi_ep_moe_requant_words.restype = ctypes.c_size_t
    L.mi_ep_moe_rowquant.argtypes = [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp]
    XCDS = int(os.environ.get("XCDS", L.mi_ep_moe_probe_xcds(None)))
    print("xcds", XCDS)
cases = [("gemm1", H, I2, cum_u), ("gemm2", I2 // 2, H, cum_u), ("gemm1 ragged", H, I2, cum_r), ("gemm2 ragged", I2 // 2, H, cum_r)]
if has_q:
    cases += [("gemm1+rowquant", H, I2, cum_u), ("gemm1q", H, I2, cum_u), ("gemm1+rowquant ragged", H, I2, cum_r), ("gemm1q ragged", H, I2, cum_r)]
for name, K, N, cum in cases:
    a = torch.randint(-8, 8, (M, K), dtype=torch.int8, device="cuda")
    w = torch.randint(-8, 8, (E, N, K), dtype=torch.int8, device="cuda")
    ws = torch.rand((E, N), device="cuda")
    if name.startswith("gemm1q"):
        q = torch.zeros((M, N // 2), dtype=torch.int8, device="cuda")
        qs = torch.zeros(M, device="cuda")
        words = torch.zeros(L.mi_ep_moe_requant_words(M, E), dtype=torch.int32, device="cuda")
        status = torch.zeros(4, dtype=torch.int32, device="cuda")
        def f():
            words.zero_()
            return L.mi_ep_moe_gemm1_swiglu_quant(ptr(a), None, ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, K, N, ptr(q), ptr(qs), ptr(words), XCDS,
                                                  ptr(status), 5000, 0, stream_ptr())
    elif name.startswith("gemm1+rowquant"):
        out = torch.zeros((M, N // 2), dtype=torch.float32, device="cuda")
        q = torch.zeros((M, N // 2), dtype=torch.int8, device="cuda")
        qs = torch.zeros(M, device="cuda")
        tot = torch.tensor([M], dtype=torch.int32, device="cuda")
        def f():
            L.mi_ep_moe_gemm1_swiglu(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, K, N, ptr(out), 0, stream_ptr())
            return L.mi_ep_moe_rowquant(ptr(out), ptr(tot), M, N // 2, ptr(q), ptr(qs), stream_ptr())
    elif name.startswith("gemm1"):
        out = torch.zeros((M, N // 2), dtype=torch.float32, device="cuda")
        f = lambda: L.mi_ep_moe_gemm1_swiglu(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, K, N, ptr(out), 0, stream_ptr())
    else:
        out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        f = lambda: L.mi_ep_moe_gemm2(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, K, N, ptr(out), 0, stream_ptr())
    for _ in range(20): assert f() == 0
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20 * 1e3)
    print(f"{name}: {best:.1f} us  {2 * M * K * N / best / 1e6:.0f} TOPS", flush=True)
    if hasattr(L, "mi_ep_moe_gemm_clock"):
        ghz, us = (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
        L.mi_ep_moe_gemm_clock(ghz, us)
        m = 0 if name.startswith("gemm1") else 1
        print(f"   shader clock under the kernel: {ghz[m]:.2f} GHz (first workgroup ran {us[m]:.0f} us)", flush=True)
    if hasattr(L, "mi_ep_gemm_dbg"):          # only in -DGEMM_TIMING builds
        import numpy as np
        buf = np.zeros(256, dtype=np.float32)
        L.mi_ep_gemm_dbg(buf.ctypes.data_as(c_vp))
        dbg = torch.from_numpy(buf).reshape(-1, 4)[:64]
        print("   cycles: per k-tile [wait, compute], per tile epilogue [until all stores issued, until drained]:", dbg.mean(dim=0).tolist())
