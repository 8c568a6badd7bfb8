"""ctypes view of the C-ABI libraries (include/mi_ep.h, include/mi_sgl_kernels.h) for the tests.
The parity tests call the HIP path through exactly these exported symbols."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "sgl-kernel-npu_amd", "lib")


def declared_symbols(header):
    """Function names declared in an include/*.h header."""
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", txt)))


def load(name):
    path = os.path.join(LIBDIR, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `python __graft_entry__.py build` (no CPU fallback exists)")
    return ctypes.CDLL(path)


_P = ctypes.c_void_p


def ptr(t):
    return _P(0) if t is None else _P(t.data_ptr())


def ptr_array(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))(*[int(p) for p in ptrs])
    return arr


def stream_ptr():
    import torch

    return _P(torch.cuda.current_stream().cuda_stream)
