"""Locates and imports the in-tree native runtime `deep_ep_cpp` (built by sgl-kernel-npu_amd/build.py).

There is deliberately no Python/CPU fallback: if the extension is missing or no AMD GPU is visible the import /
constructor fails loudly."""
import importlib
import os
import sys

_LIB = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "lib"))


def load_native():
    if "deep_ep_cpp" in sys.modules:
        return sys.modules["deep_ep_cpp"]
    import torch  # noqa: F401  (libtorch / libamdhip64 must be loaded first)

    if _LIB not in sys.path:
        sys.path.insert(0, _LIB)
    try:
        return importlib.import_module("deep_ep_cpp")
    except ImportError as e:
        raise ImportError(
            f"deep_ep_cpp (MI355X native runtime) not found under {_LIB}: run `python sgl-kernel-npu_amd/build.py` "
            f"(or `python __graft_entry__.py build`). There is no CPU fallback. Original error: {e}"
        ) from e
