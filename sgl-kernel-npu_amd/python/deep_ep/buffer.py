"""`deep_ep.Buffer` for MI355X: the public expert-parallel communication object SGLang drives.

Signature-for-signature compatible with the reference class (python/deep_ep/deep_ep/buffer.py:26-871):
constructor (:30-43), static helpers (:128-215), get_dispatch_layout (:218), dispatch (:287), combine (:473),
low_latency_dispatch (:610), low_latency_combine (:693), begin/end_profile (:741-754), fused_deep_moe (:756).
Differences that are deliberate and documented in DESIGN.md:
  * the HCCL comm-name lookup is replaced by an all-gather of hipIpc window handles over the ProcessGroup;
  * the default config tables gain a world_size-1 entry (the reference asserts at W=1, buffer.py:141-153);
  * when peers' windows cannot be mapped (no xGMI peer access) both strategies fall back to `alltoall` (RCCL)."""
import os
import socket
import warnings
from enum import IntEnum
from typing import Callable, List, Optional, Tuple, Union

import torch
import torch.distributed as dist

from ._runtime import load_native
from .ep_strategy import (LowLatencyStrategy, NormalStrategy, StrategyMap, get_low_latency_strategy,
                          get_normal_strategy)
from .utils import EventOverlap, log_parameters

_native = load_native()
Config = _native.Config
EventHandle = _native.EventHandle


class FuseMode(IntEnum):
    FUSED_DEEP_MOE = 1
    DISPATCH_FFN_COMBINE = 2


class Buffer:
    num_sms: int = 20

    # factory of the native runtime; tests that exercise only the host plumbing replace it
    _runtime_factory = staticmethod(lambda *a: _native.Buffer(*a))

    def __init__(self, group: dist.ProcessGroup, num_nvl_bytes: int = 0, num_rdma_bytes: int = 0,
                 low_latency_mode: bool = False, num_qps_per_rank: int = 12,
                 allow_nvlink_for_low_latency_mode: bool = True, allow_mnnvl: bool = False,
                 normal_strategy: Union[str, NormalStrategy] = NormalStrategy.DEFAULT,
                 low_latency_strategy: Union[str, LowLatencyStrategy] = LowLatencyStrategy.DEFAULT) -> None:
        self.group = group
        self.rank = group.rank()
        self.group_size = group.size()
        self.num_nvl_bytes = num_nvl_bytes
        self.num_rdma_bytes = num_rdma_bytes
        self.low_latency_mode = low_latency_mode
        self.moe_all_to_all_group_name = ""      # HCCL concept; RCCL communicators have no name
        self.runtime = self._runtime_factory(self.rank, self.group_size, num_nvl_bytes, num_rdma_bytes, low_latency_mode,
                                             self.moe_all_to_all_group_name)
        self.p2p_available = self._map_peer_windows()

        deep_mode = os.getenv("DEEP_USE_MODE")
        if deep_mode is not None:
            normal_strategy, low_latency_strategy = StrategyMap.get_strategy(deep_mode.lower())
        if not self.p2p_available:
            if normal_strategy == NormalStrategy.DEFAULT:
                normal_strategy = NormalStrategy.ALLTOALL
            if low_latency_strategy in (LowLatencyStrategy.DEFAULT, LowLatencyStrategy.OPS):
                low_latency_strategy = LowLatencyStrategy.ALLTOALL
        self._init_normal_strategy(normal_strategy)
        self._init_low_latency_strategy(low_latency_strategy)

    # ------------------------------------------------------------------ window bootstrap
    def _map_peer_windows(self) -> bool:
        """All-gather (host, pid, device, ipc handle, window ptr) and map every peer's window.  Returns False (on every
        rank) if any rank could not map its peers; the strategies then use torch.distributed for the byte movement."""
        rt = self.runtime
        if not hasattr(rt, "get_local_ipc_handle"):
            return bool(getattr(rt, "is_available", lambda: True)())
        if self.group_size == 1:
            return True
        if os.getenv("DEEPEP_DISABLE_P2P", "0") == "1":
            return False
        fine = [None] * self.group_size
        dist.all_gather_object(fine, bool(getattr(rt, "is_window_fine_grained", lambda: True)()), group=self.group)
        if not all(fine):        # coarse-grained window (DEEPEP_WINDOW_FINEGRAINED=0): peers' stores are not kernel-visible
            warnings.warn(f"[deep_ep rank {self.rank}] window is not fine-grained; using the alltoall strategies")
            return False
        bus = rt.get_local_device_bus_id() if hasattr(rt, "get_local_device_bus_id") else str(rt.get_local_device_id())
        me = (socket.gethostname(), os.getpid(), rt.get_local_device_id(), bytes(rt.get_local_ipc_handle()),
              list(rt.get_local_window_ptrs()), bus)
        everyone = [None] * self.group_size
        dist.all_gather_object(everyone, me, group=self.group)
        # several ranks on ONE GPU (test / dry-run setups): the low-latency calls keep their three-launch forms there -- the consuming
        # launches of the two-launch forms wait for rows in many workgroups, which needs the producers to run on GPUs of their own
        if hasattr(rt, "set_ranks_share_device"):
            rt.set_ranks_share_device(len({(h[0], h[5]) for h in everyone}) < self.group_size)

        def agree(ok: bool) -> bool:
            """Every rank reports; the step counts only if it worked everywhere.  (Also the barrier between the steps: a rank
            that failed locally still takes part in every collective, so nobody is left waiting for it.)"""
            flags = [None] * self.group_size
            dist.all_gather_object(flags, bool(ok), group=self.group)
            return all(flags)

        ok = True
        try:
            if any(h[0] != me[0] for h in everyone):
                raise RuntimeError("ranks span several hosts; windows are single-node (xGMI) only")
            handles = [h[3] for h in everyone]
            local_ptrs = [h[4] if (h[1] == me[1]) else [] for h in everyone]
            rt.sync(handles, local_ptrs)
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"[deep_ep rank {self.rank}] cannot map peer windows ({e}); using the alltoall strategies")
            ok = False
        if not agree(ok):           # every rank has mapped everybody before anyone writes into a peer's window
            return False
        # one flag + one 4 KiB row round trip with every peer (write path and read-back path, checksummed): a mapping that does
        # not behave degrades to the alltoall strategies instead of corrupting tokens later
        if hasattr(rt, "self_test") and os.getenv("DEEPEP_SKIP_SELF_TEST", "0") != "1":
            try:
                ok = bool(rt.self_test(int(os.getenv("DEEPEP_SELF_TEST_TIMEOUT_MS", "10000"))))
            except Exception as e:  # noqa: BLE001
                warnings.warn(f"[deep_ep rank {self.rank}] window self-test raised ({e})")
                ok = False
            if not ok:
                warnings.warn(f"[deep_ep rank {self.rank}] window self-test failed; using the alltoall strategies")
            if not agree(ok):
                return False
            self._check_in_launch_handoff(agree)
        return True

    def _check_in_launch_handoff(self, agree) -> None:
        """Second leg of the start-up self-test: the hand-off the TWO-launch low-latency forms rest on (a tag / flag word stored behind a
        drained write-through payload, polled and read inside one running launch -- `mi_ep_selftest_inlaunch`, four rounds over both
        ping-pong halves).  If it fails on any rank, every rank keeps the low-latency calls on their three-launch forms, where kernel
        boundaries carry the ordering; the windows themselves passed the first leg, so the strategies stay as they are."""
        rt = self.runtime
        if not hasattr(rt, "self_test_in_launch"):
            return
        # test hook: the named rank raises its tags / flags without rewriting the payload from round 1 on (what a stale line looks like)
        inject = os.getenv("DEEPEP_SELF_TEST_STALE_RANK")
        skip_from = 1 if inject is not None and int(inject) == self.rank else -1
        try:
            ok = bool(rt.self_test_in_launch(int(os.getenv("DEEPEP_SELF_TEST_TIMEOUT_MS", "10000")), skip_from))
        except Exception as e:  # noqa: BLE001
            warnings.warn(f"[deep_ep rank {self.rank}] in-launch hand-off self-test raised ({e})")
            ok = False
        everywhere = agree(ok)
        rt.set_two_launch_forms(everywhere)
        if not everywhere:
            warnings.warn(f"[deep_ep rank {self.rank}] in-launch hand-off self-test failed"
                          f"{'' if ok else ' on this rank'}; low-latency dispatch / combine use their three-launch forms")

    def _init_normal_strategy(self, strategy):
        if isinstance(strategy, NormalStrategy):
            strategy = strategy.value
        self.normal_strategy = get_normal_strategy(strategy)(runtime=self.runtime, group=self.group)

    def _init_low_latency_strategy(self, strategy, comm_alg: str = "hierarchy"):
        if isinstance(strategy, LowLatencyStrategy):
            strategy = strategy.value
        kwargs = {"runtime": self.runtime, "group": self.group}
        if strategy == "ops":
            kwargs["comm_alg"] = comm_alg
        self.low_latency_strategy = get_low_latency_strategy(strategy)(**kwargs)

    # ------------------------------------------------------------------ static helpers
    @staticmethod
    def _config_table(rows):
        return {n: Config(Buffer.num_sms, *v) for n, v in rows.items()}

    @staticmethod
    def get_dispatch_config(num_ranks: int) -> Config:
        """Recommended dispatch config (DeepEP API compatibility: only num_sms % 2 == 0 is ever checked)."""
        table = Buffer._config_table({1: (6, 256, 6, 128), 2: (24, 256, 6, 128), 4: (6, 256, 6, 128), 8: (6, 256, 6, 128),
                                      16: (36, 288, 20, 128), 24: (8, 288, 32, 128), 32: (32, 288, 32, 128),
                                      64: (20, 288, 28, 128), 128: (20, 560, 32, 128), 144: (32, 720, 12, 128),
                                      160: (28, 720, 12, 128)})
        assert num_ranks in table, f"Unsupported number of EP ranks: {num_ranks}"
        return table[num_ranks]

    @staticmethod
    def get_combine_config(num_ranks: int) -> Config:
        table = Buffer._config_table({1: (4, 256, 6, 128), 2: (10, 256, 6, 128), 4: (9, 256, 6, 128), 8: (4, 256, 6, 128),
                                      16: (4, 288, 12, 128), 24: (1, 288, 8, 128), 32: (1, 288, 8, 128),
                                      64: (1, 288, 20, 128), 128: (1, 560, 12, 128), 144: (2, 720, 8, 128),
                                      160: (2, 720, 8, 128)})
        assert num_ranks in table, f"Unsupported number of EP ranks: {num_ranks}"
        return table[num_ranks]

    @staticmethod
    def set_num_sms(new_num_sms: int) -> None:
        assert new_num_sms % 2 == 0, "The SM count must be even"
        Buffer.num_sms = new_num_sms

    @staticmethod
    def capture() -> EventOverlap:
        """Capture an event on the current stream."""
        return EventOverlap(EventHandle())

    @staticmethod
    def get_low_latency_rdma_size_hint(num_max_dispatch_tokens_per_rank: int, hidden: int, num_ranks: int,
                                       num_experts: int) -> int:
        return _native.get_low_latency_rdma_size_hint(num_max_dispatch_tokens_per_rank, hidden, num_ranks, num_experts)

    # ------------------------------------------------------------------ normal mode
    def get_dispatch_layout(self, topk_idx: torch.Tensor, num_experts: int, previous_event: Optional[EventOverlap] = None,
                            async_finish: bool = False, allocate_on_comm_stream: bool = False
                            ) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor, torch.Tensor, EventOverlap]:
        """-> (num_tokens_per_rank [W] i32, None, num_tokens_per_expert [E] i32, is_token_in_rank [T,W] i32, event)."""
        return self.normal_strategy.get_dispatch_layout(topk_idx=topk_idx, num_experts=num_experts,
                                                        previous_event=previous_event, async_finish=async_finish,
                                                        allocate_on_comm_stream=allocate_on_comm_stream)

    def get_notify_send_data(self) -> torch.Tensor:
        """Test-only accessor (reference buffer.py:256-265)."""
        return self.runtime.get_notify_send_data()

    def clean_low_latency_buffer(self, num_max_dispatch_tokens_per_rank: int, hidden: int, num_experts: int) -> None:
        """API-compat no-op (reference buffer.py:267-283); epoch-tagged flags need no cleaning."""
        self.runtime.clean_low_latency_buffer(num_max_dispatch_tokens_per_rank, hidden, num_experts)

    @log_parameters(["topk_idx"])
    def dispatch(self, x: Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]], handle: Optional[Tuple] = None,
                 num_tokens_per_rank: Optional[torch.Tensor] = None, num_tokens_per_rdma_rank: Optional[torch.Tensor] = None,
                 is_token_in_rank: Optional[torch.Tensor] = None, num_tokens_per_expert: Optional[torch.Tensor] = None,
                 topk_idx: Optional[torch.Tensor] = None, topk_weights: Optional[torch.Tensor] = None,
                 expert_alignment: int = 1, num_worst_tokens: int = 0, config: Optional[Config] = None,
                 previous_event: Optional[EventOverlap] = None, async_finish: bool = False,
                 allocate_on_comm_stream: bool = False, dispatch_wait_recv_cost_stats: Optional[torch.Tensor] = None,
                 quant_mode: Optional[str] = None):
        """Dispatch tokens to the ranks owning their experts.

        Returns (recv_x | (recv_x int8, recv_x_scales f32), recv_topk_idx, recv_topk_weights,
        num_recv_tokens_per_expert_list, handle, event).  Received rows are ordered (local expert, source rank,
        source (token, k) order).  `quant_mode` in {None, "bf16", "int8"}; None + DEEP_NORMAL_MODE_USE_INT8_QUANT=1
        selects int8 (deprecated switch kept from the reference)."""
        config = self.get_dispatch_config(self.group_size) if config is None else config
        return self.normal_strategy.dispatch(
            x=x, handle=handle, num_tokens_per_rank=num_tokens_per_rank, num_tokens_per_rdma_rank=num_tokens_per_rdma_rank,
            is_token_in_rank=is_token_in_rank, num_tokens_per_expert=num_tokens_per_expert, topk_idx=topk_idx,
            topk_weights=topk_weights, expert_alignment=expert_alignment, num_worst_tokens=num_worst_tokens, config=config,
            previous_event=previous_event, async_finish=async_finish, allocate_on_comm_stream=allocate_on_comm_stream,
            dispatch_wait_recv_cost_stats=dispatch_wait_recv_cost_stats, quant_mode=quant_mode)

    @log_parameters(["topk_idx"])
    def notify_verify(self, x, handle=None, num_tokens_per_rank=None, num_tokens_per_rdma_rank=None, is_token_in_rank=None,
                      num_tokens_per_expert=None, topk_idx=None, topk_weights=None, expert_alignment: int = 1,
                      num_worst_tokens: int = 0, config=None, previous_event=None, async_finish: bool = False,
                      allocate_on_comm_stream: bool = False, dispatch_wait_recv_cost_stats=None):
        """Test-only: run the notify exchange alone and return its 9 tables (reference buffer.py:386-470)."""
        config = self.get_dispatch_config(self.group_size) if config is None else config
        if handle is not None:
            raise NotImplementedError("Optional communication handle is not supported yet.")
        assert num_tokens_per_rank is not None and is_token_in_rank is not None and num_tokens_per_expert is not None
        use_quant = os.getenv("DEEP_NORMAL_MODE_USE_INT8_QUANT") == "1"
        return self.runtime.notify_verify(x, None, topk_idx, topk_weights, num_tokens_per_rank, is_token_in_rank,
                                          num_tokens_per_expert, 0, None, None, dispatch_wait_recv_cost_stats,
                                          expert_alignment, num_worst_tokens, config, getattr(previous_event, "event", None),
                                          async_finish, allocate_on_comm_stream, use_quant)

    @log_parameters()
    def combine(self, x: torch.Tensor, handle: Tuple, topk_weights: Optional[torch.Tensor] = None, bias=None,
                config: Optional[Config] = None, previous_event: Optional[EventOverlap] = None, async_finish: bool = False,
                allocate_on_comm_stream: bool = False, combine_send_cost_stats: Optional[torch.Tensor] = None
                ) -> Tuple[torch.Tensor, Optional[torch.Tensor], EventOverlap]:
        """Send expert outputs back and reduce them: out[t] = bf16(sum_k w[t,k] * row(t,k)) in fp32, k ascending.
        Weights are the dispatch-time `topk_weights` stored in `handle` (the argument is ignored, as in the reference)."""
        config = self.get_combine_config(self.group_size) if config is None else config
        return self.normal_strategy.combine(x=x, handle=handle, topk_weights=topk_weights, bias=bias, config=config,
                                            previous_event=previous_event, async_finish=async_finish,
                                            allocate_on_comm_stream=allocate_on_comm_stream,
                                            combine_send_cost_stats=combine_send_cost_stats)

    def internode_dispatch(self, *args, **kwargs):
        """Multi-node (RDMA) path of the reference's Ascend910B build; a single MI355X xGMI node never takes it."""
        raise NotImplementedError("internode dispatch is out of scope: one xGMI node is one rdma rank")

    def internode_combine(self, *args, **kwargs):
        raise NotImplementedError("internode combine is out of scope: one xGMI node is one rdma rank")

    # ------------------------------------------------------------------ low-latency mode
    @log_parameters(["topk_idx"])
    def low_latency_dispatch(self, x: torch.Tensor, topk_idx: torch.Tensor, num_max_dispatch_tokens_per_rank: int,
                             num_experts: int, cumulative_local_expert_recv_stats: Optional[torch.Tensor] = None,
                             use_fp8: bool = True, round_scale: bool = False, use_ue8m0: bool = False, use_mxfp4: bool = False,
                             async_finish: bool = False, return_recv_hook: bool = False,
                             topk_weights: Optional[torch.Tensor] = None, quant_mode: Optional[str] = None):
        """-> (packed_recv_x | (packed_recv_x, scales), packed_recv_count [L] i64, handle, event, hook).
        Output capacity W * num_max_dispatch_tokens_per_rank * min(K, L) rows, packed back-to-back in
        (local expert, source rank) order; no host synchronisation."""
        if quant_mode is None:           # legacy boolean mapping (reference buffer.py:668-674)
            if use_mxfp4:
                quant_mode = "mx_fp4_e2m1"
            elif use_fp8 and use_ue8m0:
                quant_mode = "mx_fp8_e4m3"
            elif use_fp8:
                quant_mode = "int8"
        return self.low_latency_strategy.low_latency_dispatch(
            x=x, topk_idx=topk_idx, num_max_dispatch_tokens_per_rank=num_max_dispatch_tokens_per_rank,
            num_experts=num_experts, cumulative_local_expert_recv_stats=cumulative_local_expert_recv_stats, use_fp8=use_fp8,
            round_scale=round_scale, use_ue8m0=use_ue8m0, use_mxfp4=use_mxfp4, async_finish=async_finish,
            return_recv_hook=return_recv_hook, topk_weights=topk_weights, quant_mode=quant_mode)

    @log_parameters(["topk_idx"])
    def low_latency_combine(self, x: torch.Tensor, topk_idx: torch.Tensor, topk_weights: torch.Tensor, handle: tuple,
                            zero_copy: bool = False, async_finish: bool = False, return_recv_hook: bool = False,
                            out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, EventOverlap, Callable]:
        return self.low_latency_strategy.low_latency_combine(x=x, topk_idx=topk_idx, topk_weights=topk_weights, handle=handle,
                                                             zero_copy=zero_copy, async_finish=async_finish,
                                                             return_recv_hook=return_recv_hook, out=out)

    # ------------------------------------------------------------------ profiling + fused MoE
    def begin_profile(self, num_profile_skip_launches: int, num_profile_active_launches: int,
                      profile_trace_dir: Optional[str] = "") -> None:
        self.runtime.begin_profile(num_profile_skip_launches, num_profile_active_launches, profile_trace_dir or "")

    def end_profile(self) -> None:
        self.runtime.end_profile()

    def get_profile_summary(self) -> dict:
        """MI355X extension: {kernel name: (launches, total_ms)} recorded between begin_profile / end_profile
        (HIP event pairs on the caller's stream around every kernel of the dispatch / combine chains)."""
        return {name: (int(n), float(ms)) for name, n, ms in self.runtime.get_profile_summary()}

    def fused_deep_moe(self, x: torch.Tensor, topk_idx: torch.Tensor, topk_weights: torch.Tensor,
                       gmm1_permuted_weight: torch.Tensor, gmm1_permuted_weight_scale: torch.Tensor,
                       gmm2_weight: torch.Tensor, gmm2_weight_scale: torch.Tensor, num_max_dispatch_tokens_per_rank: int,
                       num_experts: int, quant_mode: int = 1, fuse_mode: FuseMode = FuseMode.FUSED_DEEP_MOE,
                       profile_enable: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """dispatch -> INT8 grouped GEMM1 -> dequant+SwiGLU+requant -> grouped GEMM2 -> dequant -> combine."""
        topk_ids = topk_idx if topk_idx.dtype in (torch.int32, torch.int64) and topk_idx.is_contiguous() else topk_idx.int().contiguous()
        if fuse_mode == FuseMode.FUSED_DEEP_MOE:
            out, ep_recv_count = self.runtime.fused_deep_moe(x, topk_ids, gmm1_permuted_weight, gmm1_permuted_weight_scale,
                                                             gmm2_weight, gmm2_weight_scale, topk_weights,
                                                             num_max_dispatch_tokens_per_rank, num_experts, quant_mode,
                                                             profile_enable)
            return out, ep_recv_count
        if fuse_mode == FuseMode.DISPATCH_FFN_COMBINE:
            out, expert_token_nums = self.runtime.dispatch_ffn_combine(x, topk_ids, gmm1_permuted_weight,
                                                                       gmm1_permuted_weight_scale, gmm2_weight,
                                                                       gmm2_weight_scale, topk_weights,
                                                                       num_max_dispatch_tokens_per_rank, num_experts,
                                                                       quant_mode)
            return out, expert_token_nums
        raise NotImplementedError(f"Not support fuse_mode:{fuse_mode}")
