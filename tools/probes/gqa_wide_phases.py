"""Scratch: per-phase shader clocks of gqa_decode_wide_kernel (library built with tools/build_timing.sh <sfx> -DGQAW_STAMPS, copied over the in-tree
libmi_sgl_kernels.so for the run): reference shape, V a view of K, batch 128 x 4096 keys."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import numpy as np
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.attention.decode_attention import decode_gqa
lib = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_sgl_kernels.so"), mode=ctypes.RTLD_GLOBAL)
B, Hq, D, Dv, S, page = 128, 128, 288, 256, 4096, 64
nb = B * S // page
q = torch.randn((B, Hq, D), device="cuda").to(torch.bfloat16)
kc = torch.randn((nb, page, 1, D), device="cuda").to(torch.bfloat16)
bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(B, S // page)
o = torch.empty((B, Hq, Dv), device="cuda", dtype=torch.bfloat16)
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
for _ in range(300):
    decode_gqa(q, kc, kc[..., :Dv], o, lens, D ** -0.5, page, bt)
torch.cuda.synchronize()
ph = np.zeros((256, 8, 8), dtype=np.float32)
assert lib.mi_gqaw_phases(ph.ctypes.data_as(ctypes.c_void_p)) == 0
m = ph.mean(axis=(0, 1))
print("per tile and wave [barrier A, QK^T (+ DMA issue), softmax + publish, barrier B, P.V] shader clocks:", [int(v) for v in m[:5]], "sum", int(m[:5].sum()), "tiles", int(m[7]))
print("loop: %d clocks in %.1f us = %.2f GHz" % (m[5], m[6] / 100.0, m[5] / (m[6] * 10.0)))
for w in range(8):
    print("   wave", w, [int(v) for v in ph[:, w, :5].mean(axis=0)])
