"""sgl_kernel_npu for AMD Instinct MI355X: loads the native operator library, exactly like the reference package does
(python/sgl_kernel_npu/sgl_kernel_npu/__init__.py:9-15 -> torch.ops.load_library(lib/libsgl_kernel_npu.so)), after which
`torch.ops.npu.*` and the Python kernel functions of the sub-packages are available.  No Triton, no CPU fallback."""
import os

import torch

_LIB_DIR = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lib"))


def _load():
    path = os.path.join(_LIB_DIR, "libsgl_kernel_npu.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `python sgl-kernel-npu_amd/build.py` (there is no CPU fallback)")
    torch.ops.load_library(path)


_load()
__version__ = torch.ops.npu.sgl_kernel_npu_version()
