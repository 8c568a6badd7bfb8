"""Strategy names, abstract bases and the decorator registry.

API parity with the reference (python/deep_ep/deep_ep/ep_strategy.py:10-273): `NormalStrategy`, `LowLatencyStrategy`,
`VALID_QUANT_MODES`, `StrategyMap.get_strategy`, `register_*_strategy`, `get_*_strategy` keep their names and
error behaviour (ValueError for unknown names / DEEP_USE_MODE values)."""
from abc import ABC, abstractmethod
from typing import Dict, List, Type

import torch.distributed as dist


class NormalStrategy:
    DEFAULT = "default"      # one-sided window kernels over xGMI (deep_ep_cpp)
    ALLTOALL = "alltoall"    # torch.distributed all_to_all_single (RCCL) + HIP pack/unpack kernels

    @classmethod
    def get_all_strategies(cls) -> list:
        return [cls.DEFAULT, cls.ALLTOALL]


class LowLatencyStrategy:
    DEFAULT = "default"
    OPS = "ops"              # the reference maps this to torch_npu built-in ops; here it aliases `default`
    ALLTOALL = "alltoall"

    @classmethod
    def get_all_strategies(cls) -> list:
        return [cls.DEFAULT, cls.OPS, cls.ALLTOALL]


# accepted by the API; only bf16 / int8 exist on MI355X (the others are Ascend950-only in the reference as well)
VALID_QUANT_MODES = frozenset({"bf16", "int8", "mx_fp8_e4m3", "mx_fp8_e5m2", "pertoken_fp8_e4m3", "mx_fp4_e2m1"})


class StrategyMap:
    """DEEP_USE_MODE -> (normal strategy, low-latency strategy)."""

    strategy_map = {
        "default": (NormalStrategy.DEFAULT, LowLatencyStrategy.DEFAULT),
        "alltoall": (NormalStrategy.ALLTOALL, LowLatencyStrategy.ALLTOALL),
        "ops": (NormalStrategy.DEFAULT, LowLatencyStrategy.OPS),
    }

    @classmethod
    def get_strategy(cls, deep_mode: str):
        key = deep_mode.lower()
        if key not in cls.strategy_map:
            raise ValueError(f"Unsupported mode combination: DEEP_USE_MODE={deep_mode}, ")
        return cls.strategy_map[key]


class EPCommStrategy(ABC):
    def __init__(self, group: dist.ProcessGroup):
        self.group = group
        self._group_size = None
        self._rank = None

    @property
    def group_name(self) -> str:
        return ""          # RCCL communicators are not addressed by name

    @property
    def group_size(self) -> int:
        if self._group_size is None:
            self._group_size = self.group.size()
        return self._group_size

    @property
    def rank(self) -> int:
        if self._rank is None:
            self._rank = self.group.rank()
        return self._rank

    @abstractmethod
    def get_name(self) -> str: ...

    @abstractmethod
    def get_supported_modes(self) -> List[str]: ...


class NormalEPCommStrategy(EPCommStrategy):
    @abstractmethod
    def get_dispatch_layout(self, topk_idx, num_experts, previous_event=None, async_finish=False,
                            allocate_on_comm_stream=False): ...

    @abstractmethod
    def dispatch(self, x, handle, num_tokens_per_rank, num_tokens_per_rdma_rank, is_token_in_rank, num_tokens_per_expert,
                 topk_idx, topk_weights, expert_alignment, num_worst_tokens, config, previous_event, async_finish,
                 allocate_on_comm_stream, dispatch_wait_recv_cost_stats, quant_mode=None): ...

    @abstractmethod
    def combine(self, x, handle, topk_weights, bias, config, previous_event, async_finish, allocate_on_comm_stream,
                combine_send_cost_stats): ...


class LowLatencyEPCommStrategy(EPCommStrategy):
    @abstractmethod
    def low_latency_dispatch(self, x, topk_idx, num_max_dispatch_tokens_per_rank, num_experts,
                             cumulative_local_expert_recv_stats, use_fp8, round_scale, use_ue8m0, use_mxfp4, async_finish,
                             return_recv_hook, topk_weights, quant_mode): ...

    @abstractmethod
    def low_latency_combine(self, x, topk_idx, topk_weights, handle, zero_copy, async_finish, return_recv_hook, out): ...


_NORMAL: Dict[str, Type[NormalEPCommStrategy]] = {}
_LOW_LATENCY: Dict[str, Type[LowLatencyEPCommStrategy]] = {}


def register_normal_strategy(name: str):
    def deco(cls):
        _NORMAL[name] = cls
        return cls

    return deco


def register_low_latency_strategy(name: str):
    def deco(cls):
        _LOW_LATENCY[name] = cls
        return cls

    return deco


def get_normal_strategy(name: str) -> Type[NormalEPCommStrategy]:
    if name not in _NORMAL:
        raise ValueError(f"Unknown normal strategy: {name}. Available: {list(_NORMAL.keys())}")
    return _NORMAL[name]


def get_low_latency_strategy(name: str) -> Type[LowLatencyEPCommStrategy]:
    if name not in _LOW_LATENCY:
        raise ValueError(f"Unknown low latency strategy: {name}. Available: {list(_LOW_LATENCY.keys())}")
    return _LOW_LATENCY[name]
