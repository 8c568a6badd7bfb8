"""W expert-parallel ranks simulated inside ONE process on ONE GPU, driving the HIP kernels through the
C-ABI of include/mi_ep.h (ctypes).  Each simulated rank owns its windows (plain device buffers); "peer
pointers" are the other ranks' buffers, exactly what hipIpc-mapped windows are on a real 8-GPU node.
Post-kernels of all ranks are enqueued before any wait-kernel, so a single stream cannot deadlock.
Test infrastructure only."""
import ctypes
from ctypes import c_int, c_int32, c_size_t, c_uint32, c_uint64, c_void_p

import torch

from capi import load, ptr, ptr_array, stream_ptr

QUANT_NONE, QUANT_INT8, QUANT_INT8_NOEPS, QUANT_FP8_E4M3 = 0, 1, 2, 3


def _lib():
    lib = load("libmi_ep.so")
    V, I = c_void_p, c_int
    lib.mi_ep_version.restype = ctypes.c_char_p
    lib.mi_ep_dispatch_row_bytes.restype = c_size_t
    lib.mi_ep_dispatch_row_bytes.argtypes = [I, I]
    lib.mi_ep_combine_row_bytes.restype = c_size_t
    lib.mi_ep_combine_row_bytes.argtypes = [I]
    lib.mi_ep_dispatch_layout_workspace.restype = c_size_t
    lib.mi_ep_dispatch_layout_workspace.argtypes = [I, I, I]
    lib.mi_ep_dispatch_layout.argtypes = [V, I, I, I, I, I, V, V, V, V, V, V, c_size_t, V, V, V]
    lib.mi_ep_signal.argtypes = [V, I, I, c_uint64, V]
    lib.mi_ep_wait.argtypes = [V, I, c_uint64, V, I, V]
    lib.mi_ep_notify_post.argtypes = [V, I, I, I, V, I, c_uint32, V]
    lib.mi_ep_notify_wait.argtypes = [V, I, I, c_uint32, V, V, I, V]
    lib.mi_ep_notify_tables.argtypes = [V, I, I, I, I] + [V] * 10 + [V]
    lib.mi_ep_dispatch_stage.argtypes = [V, V, I, V, V, I, I, I, I, I, I, V, V]
    lib.mi_ep_dispatch_pull.argtypes = [V, V, V, I, I, I, I, I, V, V, V, V]
    lib.mi_ep_dispatch_index_offset.restype = c_size_t
    lib.mi_ep_dispatch_index_offset.argtypes = [I, I, I, c_size_t]
    lib.mi_ep_dispatch_stage_compact.argtypes = [V, V, I, V, V, I, I, I, I, I, I, V, c_size_t, V, c_size_t, V]
    lib.mi_ep_dispatch_pull_indexed.argtypes = [V, V, V, I, I, I, I, I, I, c_size_t, V, V, V, V, c_size_t, I, V]
    lib.mi_ep_dispatch_pull_local.argtypes = [V, V, I, V, V, V, I, I, I, I, I, I, I, I, V, V, V, V, V, c_size_t, V]
    lib.mi_ep_dispatch_push_slab_bytes.restype = c_size_t
    lib.mi_ep_dispatch_push_slab_bytes.argtypes = [c_size_t, I]
    lib.mi_ep_dispatch_stage_push.argtypes = [V, V, I, V, V, I, I, I, I, I, I, I, V, c_size_t, V, c_size_t, V]
    lib.mi_ep_dispatch_stage_push.restype = c_int
    lib.mi_ep_combine_push.argtypes = [V, V, V, I, I, I, V, I, c_size_t, V, c_size_t, I, V, V]
    lib.mi_ep_combine_reduce.argtypes = [V, V, I, V, V, V, I, I, I, I, V, V, c_size_t, V, V, I, I, I, V]
    lib.mi_ep_combine_pack.argtypes = [V, V, I, I, I, I, V, V, V]
    lib.mi_ep_combine_pack.restype = c_int
    lib.mi_ep_ll_dispatch_send.argtypes = [V, V, I, V, I, I, I, I, I, I, I, I, V, V, c_size_t, V]
    lib.mi_ep_ll_dispatch_layout_send.argtypes = [V, V, I, I, I, I, I, I, I, I, I, V, V, c_size_t, V, V, V, V, V, V]
    lib.mi_ep_ll_dispatch_layout_send.restype = c_int
    lib.mi_ep_ll_post_counts.argtypes = [V, V, I, I, I, c_uint32, V]
    lib.mi_ep_ll_dispatch_recv.argtypes = [V, V, c_uint32, I, I, I, I, I, I, V, V, V, V, V, I, V, I, V]
    # two-launch low-latency forms (tagged rows / row flags)
    lib.mi_ep_ll_dispatch_layout_send_tagged.argtypes = [V, V, I, I, I, I, I, I, I, I, I, V, V, c_size_t, V, V, V, V, V, V, c_size_t, V, V]
    lib.mi_ep_ll_dispatch_layout_send_tagged.restype = c_int
    lib.mi_ep_ll_wait_pack.argtypes = [V, V, c_size_t, I, I, I, I, I, I, V, V, V, V, V, I, V, V, c_size_t, V, I, I, V]
    lib.mi_ep_ll_wait_pack.restype = c_int
    lib.mi_ep_combine_push_flagged.argtypes = [V, V, V, I, I, I, V, I, c_size_t, V, c_size_t, I, V, V, c_size_t, V, V]
    lib.mi_ep_combine_push_flagged.restype = c_int
    lib.mi_ep_combine_reduce_flagged.argtypes = [V, V, I, V, I, I, I, I, V, V, c_size_t, V, V, I, I, I, V, c_size_t, V, V, I, I, V]
    lib.mi_ep_combine_reduce_flagged.restype = c_int
    lib.mi_ep_shared_expert_map.argtypes = [V, I, V, I, I, I, I, I, I, V, V, V]
    lib.mi_ep_shared_expert_map.restype = c_int
    for n in ("mi_ep_dispatch_layout mi_ep_signal mi_ep_wait mi_ep_notify_post mi_ep_notify_wait mi_ep_notify_tables "
              "mi_ep_dispatch_stage mi_ep_dispatch_pull mi_ep_dispatch_stage_compact mi_ep_dispatch_pull_indexed mi_ep_dispatch_pull_local mi_ep_combine_push mi_ep_combine_reduce mi_ep_ll_dispatch_send "
              "mi_ep_ll_post_counts mi_ep_ll_dispatch_recv").split():
        getattr(lib, n).restype = c_int
    return lib


LIB = None


def lib():
    global LIB
    if LIB is None:
        LIB = _lib()
    return LIB


def ck(rc):
    assert rc == 0, f"mi_ep call failed rc={rc}"


_SYNC = {}


def layout(topk_idx, E, W, coop=None, status=None, words=None):
    """-> dict of device tensors (A1).  coop: True = lend the two persistent sync words (one cooperative launch for > 1024 tokens),
    False = NULL (three launches); default alternates per call so that every test exercises both forms of the same function."""
    T, K = topk_idx.shape
    dev = topk_idx.device
    i32 = dict(dtype=torch.int32, device=dev)
    out = dict(num_tokens_per_rank=torch.empty(W, **i32), num_tokens_per_expert=torch.empty(E, **i32),
               is_token_in_rank=torch.empty((T, W), **i32), send_token_idx_small=torch.empty((T, K), **i32),
               send_data_offset=torch.empty(E, **i32))
    wsb = lib().mi_ep_dispatch_layout_workspace(T, K, E)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    if "words" not in _SYNC:
        _SYNC["words"] = torch.zeros(2, dtype=torch.int32, device=dev)        # zeroed ONCE: the barrier re-arms itself
        _SYNC["n"] = 0
    _SYNC["n"] += 1
    if coop is None:
        coop = bool(_SYNC["n"] & 1)
    ck(lib().mi_ep_dispatch_layout(ptr(topk_idx), int(topk_idx.dtype == torch.int32), T, K, E, W,
                                   ptr(out["num_tokens_per_rank"]), ptr(out["num_tokens_per_expert"]), ptr(out["is_token_in_rank"]),
                                   ptr(out["send_token_idx_small"]), ptr(out["send_data_offset"]), ptr(ws), wsb,
                                   (ptr(words) if words is not None else ptr(_SYNC["words"])) if coop else None, ptr(status) if status is not None else None, stream_ptr()))
    out["_ws"] = ws
    return out


class InProcEP:
    """Normal + low-latency dispatch/combine for W simulated ranks."""

    def __init__(self, W, E, max_tokens, K, H, device="cuda", compact=False, transport="pull"):
        self.W, self.E, self.L, self.K, self.H = W, E, E // W, K, H
        self.max_tokens = max_tokens
        self.compact = compact          # normal dispatch through stage_compact + pull_indexed (what the host runtime uses)
        # "push": mi_ep_dispatch_stage_push writes token rows + index entries into the DESTINATION ranks' regions (source slabs),
        # the receiver gathers locally with pull_indexed -- the host runtime's default at W > 1
        self.transport = transport
        # rows whose token lives on the expert rank itself do not go through the combine window (the host runtime's default);
        # exercised together with the push transport, the other modes send every row through the window
        self.combine_local = transport == "push"
        self.dev = torch.device(device)
        u8 = dict(dtype=torch.uint8, device=self.dev)
        rb = max(lib().mi_ep_dispatch_row_bytes(H, QUANT_NONE), lib().mi_ep_dispatch_row_bytes(H, QUANT_INT8))
        win_bytes = max(max_tokens * K, 1) * rb
        if transport == "push":         # W source slabs, each holding max_tokens rows + their K index entries (+ rounding slack)
            win_bytes = W * ((max_tokens + 1) * (rb + 8 * K) + 512)
        self.send_win = [torch.zeros(win_bytes, **u8) for _ in range(W)]
        self.comb_win = [torch.zeros(max(max_tokens * K, 1) * lib().mi_ep_combine_row_bytes(H), **u8) for _ in range(W)]
        self.ll_win = [torch.zeros(self.L * W * max_tokens * rb, **u8) for _ in range(W)]
        u64 = dict(dtype=torch.int64, device=self.dev)
        self.flags = [torch.zeros(4 * 64, **u64) for _ in range(W)]      # 4 flag groups of 64 slots
        self.notify = [torch.zeros(W * (E + 1), **u64) for _ in range(W)]
        self.ll_counts = [torch.zeros(self.L * W, **u64) for _ in range(W)]
        self.status = [torch.zeros(4, dtype=torch.int32, device=self.dev) for _ in range(W)]
        self.epoch = 0

    # ---- normal dispatch (A1 + A2 + A3) for all ranks; returns per-rank dicts
    def dispatch(self, xs, topk_idxs, quant_mode):
        W, E, L, K, H = self.W, self.E, self.L, self.K, self.H
        L_ = lib()
        self.epoch += 1
        ep = self.epoch
        st = stream_ptr()
        lay = [layout(topk_idxs[r], E, W) for r in range(W)]
        # post side of every rank first
        notify_ptrs = ptr_array([t.data_ptr() for t in self.notify])
        flag_ptrs = ptr_array([t.data_ptr() for t in self.flags])
        for r in range(W):
            T = xs[r].shape[0]
            if self.transport == "push":
                ck(L_.mi_ep_dispatch_stage_push(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32),
                                                ptr(lay[r]["send_token_idx_small"]), ptr(lay[r]["send_data_offset"]), T, K, H, E,
                                                W, r, quant_mode, ptr_array([t.data_ptr() for t in self.send_win]),
                                                self.send_win[r].numel(), None, 0, st))
            elif self.compact:
                ck(L_.mi_ep_dispatch_stage_compact(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32),
                                                   ptr(lay[r]["send_token_idx_small"]), ptr(lay[r]["send_data_offset"]), T, K,
                                                   H, E, r, quant_mode, ptr(self.send_win[r]), self.send_win[r].numel(), None, 0, st))
            else:
                ck(L_.mi_ep_dispatch_stage(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32),
                                           ptr(lay[r]["send_token_idx_small"]), ptr(lay[r]["send_data_offset"]), T, K, H, E,
                                           r, quant_mode, ptr(self.send_win[r]), st))
            ck(L_.mi_ep_notify_post(notify_ptrs, W, r, E, ptr(lay[r]["num_tokens_per_expert"]), T, ep, st))
            ck(L_.mi_ep_signal(flag_ptrs, W, r, ep, st))
        outs = []
        src_ptrs = ptr_array([t.data_ptr() for t in self.send_win])
        for r in range(W):
            i32 = dict(dtype=torch.int32, device=self.dev)
            cnt = torch.empty((W, E + 1), **i32)
            ck(L_.mi_ep_notify_wait(ptr(self.notify[r]), W, E, ep, ptr(cnt), ptr(self.status[r]), 2000, st))
            tb = dict(recv_count=torch.empty(L * W, **i32), recv_offset=torch.empty(L * W, **i32),
                      recv_tokens_per_expert=torch.empty(L, **i32), expert_global_offset=torch.empty(L, **i32),
                      srcrank_in_expert_offset=torch.empty(L * W, **i32), r_in_srcrank_offset=torch.empty(L * W, **i32),
                      total_recv_token=torch.empty(1, **i32), max_bs=torch.empty(1, **i32),
                      pull_offset=torch.empty(L * W, **i32))
            ck(L_.mi_ep_notify_tables(ptr(cnt), W, E, r, int(self.transport == "push"), ptr(tb["recv_count"]), ptr(tb["recv_offset"]),
                                      ptr(tb["recv_tokens_per_expert"]), ptr(tb["expert_global_offset"]),
                                      ptr(tb["srcrank_in_expert_offset"]), ptr(tb["r_in_srcrank_offset"]),
                                      ptr(tb["total_recv_token"]), ptr(tb["max_bs"]), ptr(tb["pull_offset"]), None, st))
            ck(L_.mi_ep_wait(ptr(self.flags[r]), W, ep, ptr(self.status[r]), 2000, st))
            R = int(tb["total_recv_token"].item())       # the host sync the reference also performs
            rows = max(R, 1)
            if quant_mode == QUANT_NONE:
                recv_x = torch.zeros((rows, H), dtype=torch.bfloat16, device=self.dev)
                recv_s = None
            else:
                recv_x = torch.zeros((rows, H), dtype=torch.int8, device=self.dev)
                recv_s = torch.zeros(rows, dtype=torch.float32, device=self.dev)
            src_idx = torch.zeros(rows * 3, **i32)
            if self.transport == "push":
                slab = L_.mi_ep_dispatch_push_slab_bytes(self.send_win[r].numel(), W)
                own = ptr_array([self.send_win[r].data_ptr() + s_ * slab for s_ in range(W)])
                # the host runtime's form: own tokens token by token (pull_local), the other sources row by row
                T_r = int(topk_idxs[r].shape[0])
                # the receive rows of this rank's own selections, as the dispatch knows them (sentinel: untouched entries)
                disp_local_row = torch.full((max(T_r * K, 1),), -7, dtype=torch.int32, device=self.dev)
                ck(L_.mi_ep_dispatch_pull_local(c_void_p(self.send_win[r].data_ptr() + r * slab), ptr(topk_idxs[r]),
                                                int(topk_idxs[r].dtype == torch.int32), ptr(lay[r]["send_token_idx_small"]),
                                                ptr(tb["recv_count"]), ptr(lay[r]["num_tokens_per_expert"]), T_r, K, H, E, W, r, quant_mode,
                                                R, ptr(recv_x), ptr(recv_s), ptr(src_idx), ptr(disp_local_row), None, 0, st))
                if W > 1:
                    ck(L_.mi_ep_dispatch_pull_indexed(own, ptr(tb["recv_count"]), ptr(tb["pull_offset"]), W, L, H, K, quant_mode,
                                                      R, slab, ptr(recv_x), ptr(recv_s), ptr(src_idx), None, 0, r, st))
            elif self.compact:
                ck(L_.mi_ep_dispatch_pull_indexed(src_ptrs, ptr(tb["recv_count"]), ptr(tb["pull_offset"]), W, L, H, K, quant_mode,
                                                  R, self.send_win[r].numel(), ptr(recv_x), ptr(recv_s), ptr(src_idx), None, 0, -1, st))
            else:
                ck(L_.mi_ep_dispatch_pull(src_ptrs, ptr(tb["recv_count"]), ptr(tb["pull_offset"]), W, L, H, quant_mode, R,
                                          ptr(recv_x), ptr(recv_s), ptr(src_idx), st))
            outs.append(dict(recv_x=recv_x, recv_x_scales=recv_s, recv_src_idx=src_idx, total=R, layout=lay[r], tables=tb,
                             cnt=cnt))
            if self.transport == "push":
                outs[-1]["local_row"] = disp_local_row
        torch.cuda.synchronize()
        for r in range(W):
            assert int(self.status[r][0].item()) == 0, f"rank {r} wait timed out: {self.status[r].tolist()}"
        return outs

    # ---- combine (A4 / A6)
    def combine(self, ys, src_idxs, totals, topk_idxs, topk_weights, dispatch_local_rows=None):
        """dispatch_local_rows: per rank, the `local_row` table mi_ep_dispatch_pull_local produced.  W == 1: the combine is then the
        reduce alone (no push, no signal / wait).  W > 1: the push still runs, and the table it builds for the own-rank rows must equal it."""
        W, E, K, H = self.W, self.E, self.K, self.H
        if dispatch_local_rows is not None and W == 1 and self.combine_local:
            T = topk_idxs[0].shape[0]
            out = torch.empty((T, H), dtype=torch.bfloat16, device=self.dev)
            if T:
                ck(lib().mi_ep_combine_reduce(ptr(self.comb_win[0]), ptr(topk_idxs[0]), int(topk_idxs[0].dtype == torch.int32),
                                              ptr(topk_weights[0]), None, None, T, K, H, E, ptr(out), None, 0, ptr(ys[0]),
                                              ptr(dispatch_local_rows[0]), int(ys[0].shape[0]), 0, 1, stream_ptr()))
            torch.cuda.synchronize()
            return [out]
        L_ = lib()
        self.epoch += 1
        ep = self.epoch
        st = stream_ptr()
        dst_ptrs = ptr_array([t.data_ptr() for t in self.comb_win])
        flag_ptrs = ptr_array([t.data_ptr() + 64 * 8 for t in self.flags])
        local_rows = [torch.full((max(topk_idxs[r].numel(), 1),), -1, dtype=torch.int32, device=self.dev) if self.combine_local else None
                      for r in range(W)]
        for r in range(W):
            ck(L_.mi_ep_combine_push(ptr(ys[r]), ptr(src_idxs[r]), None, int(totals[r]), H, K, dst_ptrs, W, self.comb_win[r].numel(),
                                     None, 0, r, ptr(local_rows[r]) if self.combine_local else None, st))
            ck(L_.mi_ep_signal(flag_ptrs, W, r, ep, st))
        outs = []
        for r in range(W):
            T = topk_idxs[r].shape[0]
            ck(L_.mi_ep_wait(c_void_p(self.flags[r].data_ptr() + 64 * 8), W, ep, ptr(self.status[r]), 2000, st))
            out = torch.empty((T, H), dtype=torch.bfloat16, device=self.dev)
            loc = self.combine_local and int(totals[r]) > 0
            ck(L_.mi_ep_combine_reduce(ptr(self.comb_win[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32),
                                       ptr(topk_weights[r]), None, None, T, K, H, E, ptr(out), None, 0,
                                       ptr(ys[r]) if loc else None, ptr(local_rows[r]) if loc else None, int(ys[r].shape[0]) if loc else 0,
                                       r, W, st))
            outs.append(out)
        torch.cuda.synchronize()
        if dispatch_local_rows is not None and self.combine_local:
            for r in range(W):                                   # where the push recorded a row, the dispatch had recorded the same one
                a, b = local_rows[r][:topk_idxs[r].numel()], dispatch_local_rows[r][:topk_idxs[r].numel()]
                own = a >= 0
                assert torch.equal(a[own], b[own]), r
                assert bool((b[~own] == -7).all()), r            # and nothing else
        return outs

    # ---- low-latency dispatch (A5)
    def ll_dispatch(self, xs, topk_idxs, quant_mode, count_type=1, fused=None):
        W, E, L, K, H, MT = self.W, self.E, self.L, self.K, self.H, self.max_tokens
        L_ = lib()
        self.epoch += 1
        ep = self.epoch
        st = stream_ptr()
        row_ptrs = ptr_array([t.data_ptr() for t in self.ll_win])
        cnt_ptrs = ptr_array([t.data_ptr() for t in self.ll_counts])
        lay = []
        # fused=True: the one-launch form (layout workgroup + send waves, mi_ep_ll_dispatch_layout_send) when the batch fits it -- its layout
        # tables must equal the stand-alone layout's, its rows are checked by the caller like any other; fused=None: every other call
        _SYNC["ll"] = _SYNC.get("ll", 0) + 1                  # alternates over the whole test session, whatever harness object is used
        fits = max(x.shape[0] for x in xs) <= 1024 and 16 * E <= 16384 and E % 2 == 0
        fused = ((_SYNC["ll"] & 1) == 0 if fused is None else bool(fused)) and fits
        for r in range(W):
            T = xs[r].shape[0]
            lay.append(layout(topk_idxs[r], E, W))
            if fused:
                i32 = dict(dtype=torch.int32, device=self.dev)
                f = dict(num_tokens_per_rank=torch.empty(W, **i32), num_tokens_per_expert=torch.empty(E, **i32),
                         is_token_in_rank=torch.empty((T, W), **i32), send_token_idx_small=torch.full((T, K), -9, **i32),
                         send_data_offset=torch.empty(E, **i32))
                ck(L_.mi_ep_ll_dispatch_layout_send(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32), T, K, H, E, W, r, MT,
                                                    quant_mode, row_ptrs, None, 0, ptr(f["num_tokens_per_rank"]), ptr(f["num_tokens_per_expert"]),
                                                    ptr(f["is_token_in_rank"]), ptr(f["send_token_idx_small"]), ptr(f["send_data_offset"]), st))
                torch.cuda.synchronize()
                for k_ in f:
                    assert torch.equal(f[k_], lay[r][k_]), (k_, r)
            else:
                ck(L_.mi_ep_ll_dispatch_send(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32),
                                             ptr(lay[r]["send_token_idx_small"]), T, K, H, E, W, r, MT, quant_mode, row_ptrs, None, 0, st))
            ck(L_.mi_ep_ll_post_counts(cnt_ptrs, ptr(lay[r]["num_tokens_per_expert"]), E, W, r, ep, st))
        outs = []
        M = W * MT * min(K, L)
        for r in range(W):
            i32 = dict(dtype=torch.int32, device=self.dev)
            if quant_mode == QUANT_NONE:
                px = torch.zeros((M, H), dtype=torch.bfloat16, device=self.dev)
                ps = None
            else:
                px = torch.zeros((M, H), dtype=torch.int8, device=self.dev)
                ps = torch.zeros(M, dtype=torch.float32, device=self.dev)
            prc = torch.zeros(L, dtype=torch.int64, device=self.dev)
            src_info = torch.zeros(max(xs[r].shape[0] * K, M * 128), **i32)
            rng = torch.zeros(L * W, **i32)
            ck(L_.mi_ep_ll_dispatch_recv(ptr(self.ll_win[r]), ptr(self.ll_counts[r]), ep, W, L, MT, H, quant_mode,
                                         count_type, ptr(px), ptr(ps), ptr(prc), ptr(src_info), ptr(rng), M,
                                         ptr(self.status[r]), 2000, st))
            outs.append(dict(packed_recv_x=px, packed_recv_x_scales=ps, packed_recv_count=prc, src_info=src_info,
                             layout_range=rng))
        torch.cuda.synchronize()
        for r in range(W):
            assert int(self.status[r][0].item()) == 0
        return outs


    # ---- the two-launch low-latency forms through the C-ABI: device-resident call counters, ping-pong halves, no exchange launch.
    # Every rank's producing launch is queued before any consuming launch (one stream): the consumers find their rows there, the waits
    # they contain are exercised by the multi-process tests.
    def _two_launch_state(self):
        if not hasattr(self, "tl"):
            W, L, K, H, MT = self.W, self.L, self.K, self.H, self.max_tokens
            u8 = dict(dtype=torch.uint8, device=self.dev)
            u64 = dict(dtype=torch.int64, device=self.dev)
            rb = max(lib().mi_ep_dispatch_row_bytes(H, QUANT_NONE), lib().mi_ep_dispatch_row_bytes(H, QUANT_INT8))
            rows_half = L * W * MT * rb
            cb = lib().mi_ep_combine_row_bytes(H)
            comb_half = max(MT * K, 1) * cb
            self.tl = dict(rows_half=rows_half, cnt_half=L * W * 8, comb_half=comb_half, flag_half=max(MT * K, 1) * 4,
                           rows=[torch.zeros(2 * rows_half, **u8) for _ in range(W)], cnts=[torch.zeros(2 * L * W, **u64) for _ in range(W)],
                           comb=[torch.zeros(2 * comb_half, **u8) for _ in range(W)],
                           flags=[torch.zeros(2 * max(MT * K, 1), dtype=torch.int32, device=self.dev) for _ in range(W)],
                           ll_ctr=[torch.zeros(1, **u64) for _ in range(W)], ll_cur=[torch.zeros(1, **u64) for _ in range(W)],
                           cb_ctr=[torch.zeros(1, **u64) for _ in range(W)], cb_cur=[torch.zeros(1, **u64) for _ in range(W)])
        return self.tl

    def ll_dispatch_tagged(self, xs, topk_idxs, quant_mode, count_type=1):
        W, E, L, K, H, MT = self.W, self.E, self.L, self.K, self.H, self.max_tokens
        L_ = lib()
        tl = self._two_launch_state()
        st = stream_ptr()
        row_ptrs = ptr_array([t.data_ptr() for t in tl["rows"]])
        cnt_ptrs = ptr_array([t.data_ptr() for t in tl["cnts"]])
        i32 = dict(dtype=torch.int32, device=self.dev)
        keep = []
        for r in range(W):
            T = xs[r].shape[0]
            f = dict(num_tokens_per_rank=torch.empty(W, **i32), num_tokens_per_expert=torch.empty(E, **i32),
                     is_token_in_rank=torch.empty((max(T, 1), W), **i32), send_token_idx_small=torch.full((max(T, 1), K), -9, **i32),
                     send_data_offset=torch.empty(E, **i32))
            ck(L_.mi_ep_ll_dispatch_layout_send_tagged(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32), T, K, H, E, W, r, MT,
                                                       quant_mode, row_ptrs, ptr(tl["ll_ctr"][r]), tl["rows_half"], ptr(f["num_tokens_per_rank"]),
                                                       ptr(f["num_tokens_per_expert"]), ptr(f["is_token_in_rank"]), ptr(f["send_token_idx_small"]),
                                                       ptr(f["send_data_offset"]), cnt_ptrs, tl["cnt_half"], ptr(tl["ll_cur"][r]), st))
            keep.append(f)
        outs = []
        M = W * MT * min(K, L)
        for r in range(W):
            if quant_mode == QUANT_NONE:
                px, ps = torch.zeros((M, H), dtype=torch.bfloat16, device=self.dev), None
            else:
                px, ps = torch.zeros((M, H), dtype=torch.int8, device=self.dev), torch.zeros(M, dtype=torch.float32, device=self.dev)
            prc = torch.zeros(L, dtype=torch.int64, device=self.dev)
            src_info = torch.zeros(max(xs[r].shape[0] * K, M * 128), **i32)
            rng = torch.zeros(L * W, **i32)
            ck(L_.mi_ep_ll_wait_pack(ptr(tl["rows"][r]), ptr(tl["cnts"][r]), tl["cnt_half"], W, L, MT, H, quant_mode, count_type, ptr(px), ptr(ps),
                                     ptr(prc), ptr(src_info), ptr(rng), M, ptr(tl["ll_cur"][r]), ptr(tl["ll_ctr"][r]), tl["rows_half"],
                                     ptr(self.status[r]), 2000, 0, st))
            outs.append(dict(packed_recv_x=px, packed_recv_x_scales=ps, packed_recv_count=prc, src_info=src_info, layout_range=rng, layout=keep[r]))
        torch.cuda.synchronize()
        for r in range(W):
            assert int(self.status[r][0].item()) == 0, self.status[r].tolist()
        calls = [int(t.item()) for t in tl["ll_ctr"]]
        assert len(set(calls)) == 1 and calls[0] == int(tl["ll_cur"][0].item()), (calls, "every rank completed its call counter")
        return outs

    def combine_flagged(self, ys, src_idxs, totals, topk_idxs, topk_weights):
        W, E, K, H = self.W, self.E, self.K, self.H
        L_ = lib()
        tl = self._two_launch_state()
        st = stream_ptr()
        dst_ptrs = ptr_array([t.data_ptr() for t in tl["comb"]])
        flag_ptrs = ptr_array([t.data_ptr() for t in tl["flags"]])
        local_rows = [torch.full((max(topk_idxs[r].numel(), 1),), -1, dtype=torch.int32, device=self.dev) for r in range(W)]
        for r in range(W):
            ck(L_.mi_ep_combine_push_flagged(ptr(ys[r]), ptr(src_idxs[r]), None, int(totals[r]), H, K, dst_ptrs, W, tl["comb_half"], ptr(tl["cb_ctr"][r]),
                                             tl["comb_half"], r, ptr(local_rows[r]), flag_ptrs, tl["flag_half"], ptr(tl["cb_cur"][r]), st))
        outs = []
        for r in range(W):
            T = topk_idxs[r].shape[0]
            out = torch.empty((T, H), dtype=torch.bfloat16, device=self.dev)
            loc = int(totals[r]) > 0
            ck(L_.mi_ep_combine_reduce_flagged(ptr(tl["comb"][r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32), ptr(topk_weights[r]), T, K, H,
                                               E, ptr(out), ptr(tl["cb_ctr"][r]), tl["comb_half"], ptr(ys[r]) if loc else None,
                                               ptr(local_rows[r]) if loc else None, int(ys[r].shape[0]) if loc else 0, r, W, ptr(tl["flags"][r]),
                                               tl["flag_half"], ptr(tl["cb_cur"][r]), ptr(self.status[r]), 2000, 0, st))
            outs.append(out)
        torch.cuda.synchronize()
        for r in range(W):
            assert int(self.status[r][0].item()) == 0, self.status[r].tolist()
        return outs


class InProcA2A:
    """The all-to-all (RCCL) transport of the `alltoall` strategies, simulated for W ranks in one process: the kernels are the
    real HIP ones through the C-ABI (stage -> tables(relative_pull=1) -> pull from per-source staging; combine_pack -> reduce
    in gather mode), the collective itself is replaced by device copies that move exactly the blocks all_to_all_single would."""

    def __init__(self, W, E, K, H, device="cuda"):
        self.W, self.E, self.L, self.K, self.H = W, E, E // W, K, H
        self.dev = torch.device(device)

    def dispatch(self, xs, topk_idxs, quant_mode):
        W, E, L, K, H = self.W, self.E, self.L, self.K, self.H
        L_ = lib()
        st = stream_ptr()
        rb = L_.mi_ep_dispatch_row_bytes(H, quant_mode)
        lay, rows, cnts = [], [], []
        for r in range(W):
            T = xs[r].shape[0]
            lay.append(layout(topk_idxs[r], E, W))
            buf = torch.zeros(max(T * K, 1) * rb, dtype=torch.uint8, device=self.dev)
            ck(L_.mi_ep_dispatch_stage(ptr(xs[r]), ptr(topk_idxs[r]), int(topk_idxs[r].dtype == torch.int32),
                                       ptr(lay[r]["send_token_idx_small"]), ptr(lay[r]["send_data_offset"]), T, K, H, E, r,
                                       quant_mode, ptr(buf), st))
            rows.append(buf)
            cnts.append(torch.cat([lay[r]["num_tokens_per_expert"], torch.tensor([T], dtype=torch.int32, device=self.dev)]))
        cnt = torch.stack(cnts).contiguous()                       # == all_gather_into_tensor
        ch = cnt.cpu().numpy()
        outs = []
        for me in range(W):
            i32 = dict(dtype=torch.int32, device=self.dev)
            tb = [torch.empty(max(E, 1), **i32) for _ in range(9)]
            ck(L_.mi_ep_notify_tables(ptr(cnt), W, E, me, 1, *[ptr(t) for t in tb], None, st))
            recv_count, pull_off = tb[0], tb[8]
            recv_rows = [int(ch[src, me * L:(me + 1) * L].sum()) for src in range(W)]
            R = sum(recv_rows)
            staging = torch.zeros(max(R, 1) * rb, dtype=torch.uint8, device=self.dev)
            off = 0
            bases = []
            for src in range(W):                                   # == all_to_all_single with uneven row splits
                so = int(ch[src, :me * L].sum())
                n = recv_rows[src]
                staging[off * rb:(off + n) * rb] = rows[src][so * rb:(so + n) * rb]
                bases.append(staging.data_ptr() + off * rb)
                off += n
            if quant_mode == QUANT_NONE:
                rx, rs = torch.zeros((max(R, 1), H), dtype=torch.bfloat16, device=self.dev), None
            else:
                rx = torch.zeros((max(R, 1), H), dtype=torch.int8, device=self.dev)
                rs = torch.zeros(max(R, 1), dtype=torch.float32, device=self.dev)
            src_idx = torch.zeros(max(R, 1) * 3, **i32)
            ck(L_.mi_ep_dispatch_pull(ptr_array(bases), ptr(recv_count), ptr(pull_off), W, L, H, quant_mode, R, ptr(rx), ptr(rs),
                                      ptr(src_idx), st))
            outs.append(dict(recv_x=rx, recv_x_scales=rs, recv_src_idx=src_idx, total=R, send_head=recv_count, layout=lay[me],
                             _keep=staging))
        torch.cuda.synchronize()
        self._cnt_host = ch
        return outs

    def combine(self, ys, disp, topk_idxs, topk_weights):
        W, E, L, K, H = self.W, self.E, self.L, self.K, self.H
        L_ = lib()
        st = stream_ptr()
        ch = self._cnt_host
        packed, per_src = [], []
        for r in range(W):
            pk = torch.zeros_like(ys[r])
            rps = torch.zeros(W, dtype=torch.int32, device=self.dev)
            ck(L_.mi_ep_combine_pack(ptr(ys[r]), ptr(disp[r]["send_head"]), W, L, H, ys[r].shape[0], ptr(pk), ptr(rps), st))
            packed.append(pk)
            per_src.append(rps)
        torch.cuda.synchronize()
        outs = []
        for me in range(W):
            T = topk_idxs[me].shape[0]
            n_pairs = int(ch[me, :E].sum())
            ret = torch.zeros((max(n_pairs, 1), H), dtype=torch.bfloat16, device=self.dev)
            off = 0
            for d in range(W):                                     # block d of my return buffer comes from expert rank d
                n = int(ch[me, d * L:(d + 1) * L].sum())
                rps = per_src[d].cpu().numpy()
                so = int(rps[:me].sum())
                assert int(rps[me]) == n
                ret[off:off + n] = packed[d][so:so + n]
                off += n
            out = torch.empty((T, H), dtype=torch.bfloat16, device=self.dev)
            lay = disp[me]["layout"]
            ck(L_.mi_ep_combine_reduce(ptr(ret), ptr(topk_idxs[me]), int(topk_idxs[me].dtype == torch.int32), ptr(topk_weights[me]),
                                       ptr(lay["send_data_offset"]), ptr(lay["send_token_idx_small"]), T, K, H, E, ptr(out), None, 0,
                                       None, None, 0, 0, 1, st))
            outs.append(out)
        torch.cuda.synchronize()
        return outs
