"""Normal-mode (high-throughput) dispatch / combine strategies for MI355X.

`default`  -> deep_ep_cpp window kernels (one-sided over xGMI), mirrors DefaultNormalCommStrategy of the reference
              (python/deep_ep/deep_ep/strategies/normal_strategy.py:25-459): same quant-mode selection (:163-206),
              same handle 8-tuple (:253-262) and return arity (:264-271).
`alltoall` -> torch.distributed all_to_all_single (RCCL on GPUs, gloo in the CPU plumbing tests) moving rows that the
              HIP kernels pack / unpack; same algorithm shape as the reference's AlltoAllNormalCommStrategy (:462-849:
              counts all-gather -> permute -> all_to_all_single with uneven splits -> un-permute + weighted sum) with the
              torch_npu routing ops replaced by our kernels.  Results are identical to `default` (same kernels, same
              row order)."""
import os
from typing import List, Optional, Tuple, Union

import torch
import torch.distributed as dist

from ..ep_strategy import VALID_QUANT_MODES, NormalEPCommStrategy, register_normal_strategy
from ..utils import EventOverlap


def resolve_quant(x, quant_mode: Optional[str]):
    """-> (data tensor, quant_type, use_quant); reference normal_strategy.py:163-206."""
    if quant_mode is None:
        if isinstance(x, torch.Tensor):
            data, quant_type, use_quant = x, "bf16", False
        elif isinstance(x, tuple) and len(x) == 2:
            data, tag = x
            names = {torch.float8_e4m3fn: "mx_fp8_e4m3", torch.float8_e5m2: "mx_fp8_e5m2", torch.int8: "int8"}
            if hasattr(torch, "float4_e2m1fn_x2"):
                names[torch.float4_e2m1fn_x2] = "mx_fp4_e2m1"
            if tag.dtype not in names:
                raise TypeError(f"Unsupported quantized dtype: {tag.dtype}")
            quant_type, use_quant = names[tag.dtype], True
        else:
            raise TypeError(f"Unsupported x type: {type(x)}")
        if not use_quant and os.getenv("DEEP_NORMAL_MODE_USE_INT8_QUANT") == "1":     # deprecated switch
            quant_type, use_quant = "int8", True
        return data, quant_type, use_quant
    if quant_mode not in VALID_QUANT_MODES:
        raise ValueError(f"Invalid quant_mode: {quant_mode}. Valid options: {VALID_QUANT_MODES}")
    return x, quant_mode, quant_mode != "bf16"


def _event(e):
    return getattr(e, "event", None)


def long_seq_rounds():
    """(rounds, tokens per round) of the reference's long-sequence mode (csrc/deepep/deep_ep.cpp:63-90, README.md:224-226):
    DEEPEP_NORMAL_LONG_SEQ_ROUND in [1, 256], DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS in [32, 8192], product <= 131072."""
    rounds = int(os.getenv("DEEPEP_NORMAL_LONG_SEQ_ROUND", "1"))
    per = int(os.getenv("DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS", "8192"))
    if not (1 <= rounds <= 256):
        raise ValueError(f"DEEPEP_NORMAL_LONG_SEQ_ROUND ({rounds}) must be in [1, 256]")
    if not (32 <= per <= 8192):
        raise ValueError(f"DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS ({per}) must be in [32, 8192]")
    if rounds * per > 131072:
        raise ValueError(f"DEEPEP_NORMAL_LONG_SEQ_ROUND ({rounds}) * DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS ({per}) must not exceed 131072")
    return rounds, per


@register_normal_strategy("default")
class DefaultNormalCommStrategy(NormalEPCommStrategy):
    def __init__(self, runtime, group: dist.ProcessGroup):
        super().__init__(group)
        self.runtime = runtime

    def get_name(self) -> str:
        return "default"

    def get_supported_modes(self) -> List[str]:
        return ["normal"]

    def get_dispatch_layout(self, topk_idx, num_experts, previous_event=None, async_finish=False,
                            allocate_on_comm_stream=False):
        self.num_experts = num_experts
        per_rank, per_rdma_rank, per_expert, is_in_rank, event = self.runtime.get_dispatch_layout(
            topk_idx, num_experts, _event(previous_event), async_finish, allocate_on_comm_stream)
        return per_rank, per_rdma_rank, per_expert, is_in_rank, EventOverlap(event)

    def dispatch(self, x, handle, num_tokens_per_rank, num_tokens_per_rdma_rank, is_token_in_rank, num_tokens_per_expert,
                 topk_idx, topk_weights, expert_alignment=1, num_worst_tokens=0, config=None, previous_event=None,
                 async_finish=False, allocate_on_comm_stream=False, dispatch_wait_recv_cost_stats=None, quant_mode=None):
        data, quant_type, use_quant = resolve_quant(x, quant_mode)
        if handle is not None:
            raise NotImplementedError("Optional communication handle is not supported yet.")
        assert num_tokens_per_rank is not None and is_token_in_rank is not None and num_tokens_per_expert is not None
        rounds, per_round = long_seq_rounds()
        if rounds > 1:
            assert num_worst_tokens == 0, "num_worst_tokens is not supported together with DEEPEP_NORMAL_LONG_SEQ_ROUND > 1"
            return self._dispatch_rounds(data, quant_type, use_quant, is_token_in_rank, num_tokens_per_expert, topk_idx,
                                         topk_weights, expert_alignment, config, dispatch_wait_recv_cost_stats, rounds, per_round)
        (recv_x, recv_x_scales, recv_topk_idx, recv_topk_weights, num_recv_tokens_per_expert_list, rank_prefix_matrix,
         channel_prefix_matrix, recv_channel_prefix_matrix, recv_src_idx, send_head, event) = self.runtime.intranode_dispatch(
            data, None, topk_idx, topk_weights, num_tokens_per_rank, is_token_in_rank, num_tokens_per_expert, 0, None, None,
            dispatch_wait_recv_cost_stats, expert_alignment, num_worst_tokens, config, _event(previous_event), async_finish,
            allocate_on_comm_stream, use_quant, quant_type)
        handle = (rank_prefix_matrix, channel_prefix_matrix, recv_channel_prefix_matrix, recv_src_idx, is_token_in_rank,
                  send_head, topk_idx, topk_weights)
        return ((recv_x, recv_x_scales) if use_quant else recv_x, recv_topk_idx, recv_topk_weights,
                num_recv_tokens_per_expert_list, handle, EventOverlap(event))

    # ---- long-sequence mode (reference "ant moving home": deep_ep.cpp:63-90, round loop
    # cam_moe_dispatch_normal.h:767-781).  The reference re-uses a fixed-size window by sending `rounds` slices of
    # `per_round` tokens through ONE kernel; here every slice is a complete exchange through the tested single-shot path
    # (so the window only ever holds one slice) and the receive side is re-assembled into the single-shot order -- rows
    # sorted by (local expert, source rank, token) -- with index copies.  Every rank must set the same two env values.
    def _dispatch_rounds(self, data, quant_type, use_quant, is_token_in_rank, num_tokens_per_expert, topk_idx, topk_weights,
                         expert_alignment, config, stats, rounds, per_round):
        T, E = topk_idx.size(0), num_tokens_per_expert.numel()
        if T > rounds * per_round:
            raise ValueError(f"{T} tokens exceed DEEPEP_NORMAL_LONG_SEQ_ROUND * PER_ROUND_TOKENS = {rounds * per_round}")
        dev = data.device
        chunks, lists = [], []
        for r in range(rounds):
            t0, t1 = min(T, r * per_round), min(T, (r + 1) * per_round)
            ti, wi = topk_idx[t0:t1].contiguous(), topk_weights[t0:t1].contiguous()
            per_rank, _, per_expert, is_in, _ = self.runtime.get_dispatch_layout(ti, E, None, False, False)
            (rx, rs, _, _, lst, rpm, cpm, rcpm, src_idx, send_head, _) = self.runtime.intranode_dispatch(
                data[t0:t1].contiguous(), None, ti, wi, per_rank, is_in, per_expert, 0, None, None, stats, expert_alignment, 0,
                config, None, False, False, use_quant, quant_type)
            cum = send_head.to(torch.int64)
            cnt = cum - torch.cat([cum.new_zeros(1), cum[:-1]])
            # rows received in this slice, from the host-side per-expert list (counts, or inclusive sums when
            # MOE_EXPERT_TOKEN_NUMS_TYPE=0)
            cumulative = int(os.getenv("MOE_EXPERT_TOKEN_NUMS_TYPE", "1")) == 0
            n = (int(lst[-1]) if cumulative else int(sum(lst))) if lst else 0
            chunks.append(dict(t0=t0, n=n, rx=rx[:n], rs=rs[:n], src=src_idx[:3 * n].view(n, 3), cnt=cnt, src_idx=src_idx,
                               send_head=send_head, topk_idx=ti, topk_weights=wi))
            lists.append(lst)
        total_cnt = sum(c["cnt"] for c in chunks)
        final_cum = torch.cumsum(total_cnt, 0)
        final_excl = final_cum - total_cnt
        N = sum(c["n"] for c in chunks)
        rows = max(N, 1)
        recv_x = torch.empty((rows,) + tuple(chunks[0]["rx"].shape[1:]), dtype=chunks[0]["rx"].dtype, device=dev)
        recv_scales = torch.empty((rows,), dtype=torch.float32, device=dev)
        recv_src = torch.empty((rows, 3), dtype=torch.int32, device=dev)
        running = torch.zeros_like(total_cnt)
        seg_ids = torch.arange(E, device=dev)
        for c in chunks:
            n = c["n"]
            seg = torch.repeat_interleave(seg_ids, c["cnt"], output_size=n)
            excl = torch.cumsum(c["cnt"], 0) - c["cnt"]
            dest = final_excl[seg] + running[seg] + (torch.arange(n, device=dev) - excl[seg])
            running = running + c["cnt"]
            recv_x.index_copy_(0, dest, c["rx"])
            recv_scales.index_copy_(0, dest, c["rs"])
            src = c["src"].clone()
            src[:, 1] += c["t0"]                               # token index inside the whole batch
            recv_src.index_copy_(0, dest, src)
            c["dest"] = dest
            del c["rx"], c["rs"], c["src"], c["cnt"]
        counts = [sum(v) for v in zip(*lists)] if lists and lists[0] else []
        final_src_idx = recv_src.reshape(-1)
        final_src_idx._mi_long_seq = chunks                      # per-slice handles for combine()
        handle = (rpm, cpm, rcpm, final_src_idx, is_token_in_rank, final_cum.to(torch.int32), topk_idx, topk_weights)
        recv_topk_idx = torch.empty((N, topk_idx.size(1)), dtype=topk_idx.dtype, device=dev)
        recv_topk_weights = torch.empty((N, topk_idx.size(1)), dtype=topk_weights.dtype, device=dev)
        return ((recv_x, recv_scales) if use_quant else recv_x, recv_topk_idx, recv_topk_weights, counts, handle,
                EventOverlap(None))

    def combine(self, x, handle, topk_weights=None, bias=None, config=None, previous_event=None, async_finish=False,
                allocate_on_comm_stream=False, combine_send_cost_stats=None):
        # weights come from the handle (dispatch-time topk_weights); the `topk_weights` argument is ignored, exactly
        # like the reference (normal_strategy.py:407-420)
        _, _, _, src_idx, _, send_head, topk_idx, topk_weights_ori = handle
        chunks = getattr(src_idx, "_mi_long_seq", None)
        if chunks is not None:                                   # long-sequence mode: slice by slice, same order as dispatch
            outs = []
            for c in chunks:
                out_r, _, _ = self.runtime.intranode_combine(x.index_select(0, c["dest"]), c["topk_idx"], c["topk_weights"],
                                                             c["src_idx"], c["send_head"], combine_send_cost_stats)
                outs.append(out_r)
            return torch.cat(outs, dim=0), None, EventOverlap(None)
        recv_x, recv_topk_weights, event = self.runtime.intranode_combine(x, topk_idx, topk_weights_ori, src_idx, send_head,
                                                                         combine_send_cost_stats)
        return recv_x, recv_topk_weights, EventOverlap(event)


@register_normal_strategy("alltoall")
class AlltoAllNormalCommStrategy(NormalEPCommStrategy):
    """Rows travel with torch.distributed; the runtime only packs / unpacks (see module docstring)."""

    def __init__(self, runtime, group: dist.ProcessGroup):
        super().__init__(group)
        self.runtime = runtime
        self.num_experts = None

    def get_name(self) -> str:
        return "alltoall"

    def get_supported_modes(self) -> List[str]:
        return ["normal"]

    def get_dispatch_layout(self, topk_idx, num_experts, previous_event=None, async_finish=False,
                            allocate_on_comm_stream=False):
        self.num_experts = num_experts
        per_rank, per_rdma_rank, per_expert, is_in_rank, event = self.runtime.get_dispatch_layout(
            topk_idx, num_experts, _event(previous_event), async_finish, allocate_on_comm_stream)
        return per_rank, per_rdma_rank, per_expert, is_in_rank, EventOverlap(event)

    # -- collectives (thin wrappers so the byte movement is in one place)
    def _all_gather_counts(self, vec: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self.group_size, vec.numel()), dtype=vec.dtype, device=vec.device)
        dist.all_gather_into_tensor(out.view(-1), vec.contiguous(), group=self.group)
        return out

    def _all_to_all_rows(self, send: torch.Tensor, send_rows: List[int], recv_rows: List[int]) -> torch.Tensor:
        """send [n, row_bytes] uint8/bf16, row-granular uneven splits."""
        out = torch.empty((max(sum(recv_rows), 1),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        n_send = sum(send_rows)
        dist.all_to_all_single(out[:sum(recv_rows)], send[:n_send], output_split_sizes=list(recv_rows),
                               input_split_sizes=list(send_rows), group=self.group)
        return out

    def dispatch(self, x, handle, num_tokens_per_rank, num_tokens_per_rdma_rank, is_token_in_rank, num_tokens_per_expert,
                 topk_idx, topk_weights, expert_alignment=1, num_worst_tokens=0, config=None, previous_event=None,
                 async_finish=False, allocate_on_comm_stream=False, dispatch_wait_recv_cost_stats=None, quant_mode=None):
        data, quant_type, use_quant = resolve_quant(x, quant_mode)
        if quant_type not in ("bf16", "int8", "pertoken_fp8_e4m3"):
            raise ValueError(f"{quant_type} is not supported on this device, please use int8 or bf16 instead.")
        if handle is not None:
            raise NotImplementedError("Optional communication handle is not supported yet.")
        assert num_tokens_per_rank is not None and is_token_in_rank is not None and num_tokens_per_expert is not None
        num_experts = int(num_tokens_per_expert.numel())
        hidden = int(data.size(1))
        rows, cnt_vec = self.runtime.a2a_dispatch_stage(data, topk_idx, num_experts, quant_type)
        cnt_matrix = self._all_gather_counts(cnt_vec)
        (recv_count, pull_offset, send_rows, recv_rows, per_expert, total_recv, _max_bs) = \
            self.runtime.a2a_dispatch_tables(cnt_matrix)
        staging = self._all_to_all_rows(rows, send_rows, recv_rows)
        recv_x, recv_x_scales, recv_src_idx = self.runtime.a2a_dispatch_unpack(
            staging, recv_rows, recv_count, pull_offset, hidden, total_recv, quant_type, 0, 0)
        if os.getenv("MOE_EXPERT_TOKEN_NUMS_TYPE", "1") == "0":
            run, cum = 0, []
            for c in per_expert:
                run += c
                cum.append(run)
            per_expert = cum
        W = self.group_size
        i32 = dict(dtype=torch.int32, device=data.device)
        channels = max((config.num_sms if config is not None else 20) // 2, 1)
        handle = (torch.zeros((W, W), **i32), torch.zeros((W, channels), **i32), torch.zeros((W, channels), **i32),
                  recv_src_idx, is_token_in_rank, recv_count, topk_idx, topk_weights)
        recv_topk_idx = torch.empty((total_recv, topk_idx.size(1)), dtype=topk_idx.dtype, device=data.device)
        recv_topk_weights = torch.empty((total_recv, topk_idx.size(1)), dtype=torch.float32, device=data.device)
        return ((recv_x, recv_x_scales) if use_quant else recv_x, recv_topk_idx, recv_topk_weights, list(per_expert),
                handle, EventOverlap(None))

    def combine(self, x, handle, topk_weights=None, bias=None, config=None, previous_event=None, async_finish=False,
                allocate_on_comm_stream=False, combine_send_cost_stats=None):
        _, _, _, _src_idx, _, send_head, topk_idx, topk_weights_ori = handle
        num_experts = int(send_head.numel())
        hidden = int(x.size(1))
        packed, rows_per_src = self.runtime.a2a_combine_pack(x, send_head)
        send_off, idx_small, rows_sent = self.runtime.a2a_combine_prepare(topk_idx, num_experts)
        returned = self._all_to_all_rows(packed, rows_per_src, rows_sent)
        out = self.runtime.a2a_combine_reduce(returned, topk_idx, topk_weights_ori, send_off, idx_small, hidden, num_experts)
        return out, None, EventOverlap(None)
