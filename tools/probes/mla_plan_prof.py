"""Scratch (run under rocprofv3 --kernel-trace --stats): one ragged batch, planned form then AB_N uniform splits, 60 calls each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.bench_hooks import _mla_inputs

B, Hq, S, page = int(os.environ.get("AB_B", 64)), int(os.environ.get("AB_HQ", 128)), int(os.environ.get("AB_S", 8192)), 64
q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
_, _, _, _, rlens = _mla_inputs(B, Hq, S, page, ragged=True)
out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
for n in (0, int(os.environ.get("AB_N", 4))):
    for _ in range(60):
        torch.ops.npu.decode_mla(q, kn, kr, out, rlens, 576 ** -0.5, page, bt, n)
    torch.cuda.synchronize()
