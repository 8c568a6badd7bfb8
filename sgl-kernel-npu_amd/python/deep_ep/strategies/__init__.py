"""Strategy implementations (importing this package registers them)."""
from ..ep_strategy import (EPCommStrategy, LowLatencyEPCommStrategy, NormalEPCommStrategy, get_low_latency_strategy,
                           get_normal_strategy, register_low_latency_strategy, register_normal_strategy)
from .low_latency_strategy import (AllToAllLowLatencyCommStrategy, DefaultLowLatencyCommStrategy,
                                   OpsLowLatencyCommStrategy)
from .normal_strategy import AlltoAllNormalCommStrategy, DefaultNormalCommStrategy

__all__ = [
    "EPCommStrategy", "NormalEPCommStrategy", "LowLatencyEPCommStrategy", "register_normal_strategy",
    "register_low_latency_strategy", "get_normal_strategy", "get_low_latency_strategy", "DefaultNormalCommStrategy",
    "AlltoAllNormalCommStrategy", "DefaultLowLatencyCommStrategy", "OpsLowLatencyCommStrategy",
    "AllToAllLowLatencyCommStrategy",
]
