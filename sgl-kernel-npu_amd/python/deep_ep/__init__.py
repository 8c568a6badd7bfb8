"""deep_ep for AMD Instinct MI355X (gfx950) -- drop-in for sgl-kernel-npu's `deep_ep` package
(reference python/deep_ep/deep_ep/__init__.py:14-20 exports the same names)."""
from ._runtime import load_native as _load_native

_native = _load_native()
Config = _native.Config

from . import strategies  # noqa: E402,F401  (registers the strategies)
from .buffer import Buffer  # noqa: E402
from .ep_strategy import LowLatencyStrategy, NormalStrategy  # noqa: E402
from .utils import EventOverlap  # noqa: E402

__all__ = ["Buffer", "Config", "EventOverlap", "NormalStrategy", "LowLatencyStrategy"]
