"""Scratch: BASELINE C4 MLA decode in a loop with one kernel form (argv[1] = 4 | 8), for rocprofv3 passes."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
import sgl_kernel_npu
lib = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_sgl_kernels.so"), mode=ctypes.RTLD_GLOBAL)
lib.mi_mla_decode_select_wide(int(sys.argv[1]))
q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64, ragged=False)
out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device="cuda")
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, 0)
torch.cuda.synchronize()
