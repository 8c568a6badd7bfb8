"""Low-latency dispatch / combine strategies for MI355X.

`default` (and its alias `ops`) -> deep_ep_cpp window kernels: rows are written straight into the destination rank's
window, no host sync, worst-case sized outputs; mirrors DefaultLowLatencyCommStrategy of the reference
(python/deep_ep/deep_ep/strategies/low_latency_strategy.py:18-177): same handle 7-tuple (:93-101) and return arity.
`alltoall` -> torch.distributed transport with the same packing kernels (reference :459-639 uses dist.all_to_all)."""
from typing import List, Optional

import torch
import torch.distributed as dist

from ..ep_strategy import LowLatencyEPCommStrategy, register_low_latency_strategy
from ..utils import EventOverlap

_VALID_LL_QUANT = {None, "int8", "mx_fp8_e4m3", "mx_fp8_e5m2", "pertoken_fp8_e4m3", "mx_fp4_e2m1"}


def _as_index(topk_idx: torch.Tensor) -> torch.Tensor:
    """int32 / int64 contiguous indices go to the runtime as they are; anything else is narrowed to int32 like the reference does."""
    if topk_idx.dtype in (torch.int32, torch.int64) and topk_idx.is_contiguous():
        return topk_idx
    return topk_idx.int().contiguous()



@register_low_latency_strategy("default")
class DefaultLowLatencyCommStrategy(LowLatencyEPCommStrategy):
    def __init__(self, runtime, group: dist.ProcessGroup, comm_alg: str = "hierarchy"):
        super().__init__(group)
        self.runtime = runtime

    def get_name(self) -> str:
        return "default"

    def get_supported_modes(self) -> List[str]:
        return ["low_latency"]

    def low_latency_dispatch(self, x, topk_idx, num_max_dispatch_tokens_per_rank, num_experts,
                             cumulative_local_expert_recv_stats=None, use_fp8=True, round_scale=False, use_ue8m0=False,
                             use_mxfp4=False, async_finish=False, return_recv_hook=False, topk_weights=None, quant_mode=None):
        if quant_mode not in _VALID_LL_QUANT:
            raise ValueError(f"Unsupported quant_mode: {quant_mode}")
        # (the reference narrows to int32 here, :57; the kernels below read either width, and at decode sizes the conversion is a
        # launch that costs as much as the layout kernel)
        topk_ids = _as_index(topk_idx)
        (packed_recv_x, packed_recv_x_scales, packed_recv_count, packed_recv_src_info, packed_recv_layout_range, event,
         hook) = self.runtime.low_latency_dispatch(
            x, topk_ids, cumulative_local_expert_recv_stats, num_max_dispatch_tokens_per_rank, num_experts, use_fp8,
            round_scale, use_ue8m0, use_mxfp4, async_finish, return_recv_hook, "none" if quant_mode is None else quant_mode)
        handle = (packed_recv_src_info, packed_recv_layout_range, num_max_dispatch_tokens_per_rank, x.size(1), num_experts,
                  packed_recv_count, None)
        keep = (x, topk_idx, packed_recv_x, packed_recv_x_scales, packed_recv_count, packed_recv_src_info,
                packed_recv_layout_range, cumulative_local_expert_recv_stats)
        return ((packed_recv_x, packed_recv_x_scales) if quant_mode is not None else packed_recv_x, packed_recv_count, handle,
                EventOverlap(event, keep if async_finish else None), hook)

    def low_latency_combine(self, x, topk_idx, topk_weights, handle, zero_copy=False, async_finish=False,
                            return_recv_hook=False, out=None):
        topk_ids = _as_index(topk_idx)
        src_info, layout_range, num_max_dispatch_tokens_per_rank, _hidden, num_experts, packed_recv_count, _ = handle
        combined_x, event, hook = self.runtime.low_latency_combine(
            x, topk_ids, topk_weights, src_info, layout_range, num_max_dispatch_tokens_per_rank, num_experts,
            packed_recv_count, zero_copy, async_finish, return_recv_hook, out)
        keep = (x, topk_idx, topk_weights, src_info, layout_range, combined_x)
        return combined_x, EventOverlap(event, keep if async_finish else None), hook


@register_low_latency_strategy("ops")
class OpsLowLatencyCommStrategy(DefaultLowLatencyCommStrategy):
    """The reference routes `ops` to torch_npu's built-in npu_moe_distribute_dispatch_v2 / combine_v2
    (low_latency_strategy.py:180-456).  There is no vendor op to defer to on ROCm, so `ops` is the default strategy."""

    def get_name(self) -> str:
        return "ops"


@register_low_latency_strategy("alltoall")
class AllToAllLowLatencyCommStrategy(LowLatencyEPCommStrategy):
    def __init__(self, runtime, group: dist.ProcessGroup, comm_alg: str = "hierarchy"):
        super().__init__(group)
        self.runtime = runtime

    def get_name(self) -> str:
        return "alltoall"

    def get_supported_modes(self) -> List[str]:
        return ["low_latency"]

    def _all_to_all_rows(self, send, send_rows, recv_rows):
        out = torch.empty((max(sum(recv_rows), 1),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out[:sum(recv_rows)], send[:sum(send_rows)], output_split_sizes=list(recv_rows),
                               input_split_sizes=list(send_rows), group=self.group)
        return out

    def low_latency_dispatch(self, x, topk_idx, num_max_dispatch_tokens_per_rank, num_experts,
                             cumulative_local_expert_recv_stats=None, use_fp8=True, round_scale=False, use_ue8m0=False,
                             use_mxfp4=False, async_finish=False, return_recv_hook=False, topk_weights=None, quant_mode=None):
        if quant_mode not in _VALID_LL_QUANT:
            raise ValueError(f"Unsupported quant_mode: {quant_mode}")
        if quant_mode not in (None, "int8", "pertoken_fp8_e4m3"):
            raise ValueError(f"{quant_mode} is not supported on this device, please use int8, pertoken_fp8_e4m3 or bf16 instead.")
        import os

        if int(os.getenv("MOE_SHARED_EXPERT_RANK_NUM", "0")) != 0:
            raise ValueError("MOE_SHARED_EXPERT_RANK_NUM is served by the default (window) low-latency strategy only")
        topk_ids = topk_idx.int()
        W = self.group_size
        L = num_experts // W
        K = topk_ids.size(1)
        hidden = x.size(1)
        qt = {"int8": "int8_ll", "pertoken_fp8_e4m3": "pertoken_fp8_e4m3"}.get(quant_mode, "bf16")
        rows, cnt_vec = self.runtime.a2a_dispatch_stage(x, topk_ids, num_experts, qt)
        cnt_matrix = torch.empty((W, cnt_vec.numel()), dtype=cnt_vec.dtype, device=cnt_vec.device)
        dist.all_gather_into_tensor(cnt_matrix.view(-1), cnt_vec, group=self.group)
        recv_count, pull_offset, send_rows, recv_rows, per_expert, total_recv, _ = self.runtime.a2a_dispatch_tables(cnt_matrix)
        staging = self._all_to_all_rows(rows, send_rows, recv_rows)
        M = W * num_max_dispatch_tokens_per_rank * min(K, L)
        src_len = max(x.size(0) * K, M * 128)
        packed_recv_x, scales, src_info = self.runtime.a2a_dispatch_unpack(staging, recv_rows, recv_count, pull_offset, hidden,
                                                                           total_recv, qt, M, src_len)
        counts = torch.tensor(per_expert, dtype=torch.int64)
        if os.getenv("MOE_EXPERT_TOKEN_NUMS_TYPE", "1") == "0":
            counts = counts.cumsum(0)
        packed_recv_count = counts.to(x.device)
        handle = (src_info, recv_count, num_max_dispatch_tokens_per_rank, hidden, num_experts, packed_recv_count, None)
        if quant_mode is None:
            ret_x = packed_recv_x
        else:
            ret_x = (packed_recv_x, scales)
        return ret_x, packed_recv_count, handle, EventOverlap(None), (lambda: None)

    def low_latency_combine(self, x, topk_idx, topk_weights, handle, zero_copy=False, async_finish=False,
                            return_recv_hook=False, out=None):
        topk_ids = topk_idx.int()
        _src_info, layout_range, _mt, hidden, num_experts, _cnt, _ = handle
        packed, rows_per_src = self.runtime.a2a_combine_pack(x, layout_range)
        send_off, idx_small, rows_sent = self.runtime.a2a_combine_prepare(topk_ids, num_experts)
        returned = self._all_to_all_rows(packed, rows_per_src, rows_sent)
        combined = self.runtime.a2a_combine_reduce(returned, topk_ids, topk_weights, send_off, idx_small, hidden, num_experts)
        return combined, EventOverlap(None), (lambda: None)
