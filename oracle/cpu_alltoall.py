"""CPU baseline with the shape of the real exchange: W processes over gloo running the reference's `alltoall` strategy
(python/deep_ep/deep_ep/strategies/normal_strategy.py:481-790) with the torch_npu routing ops restated in plain torch:

  get_dispatch_layout  :491-571   histc of topk_idx -> all-gather of the per-expert counts -> input / output splits,
                                  global_tokens_indices = repeat_interleave(local expert id, counts)
  dispatch             :573-727   npu_moe_init_routing_v2(x, topk_idx, quant_mode=1) = rows replicated in expert-sorted
                                  order (stable) + per-token dynamic INT8 quantisation; all_to_all of scales and tokens with
                                  uneven splits; second routing pass = stable sort of the received rows by local expert
  combine              :729-780   npu_moe_finalize_routing = undo the second sort; all_to_all back; undo the first sort with
                                  the top-k weights applied (fp32 accumulate) -> bf16

TEST / BENCHMARK INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ and by bench.py's `cpu_baseline` leg, never by
the product path.  Quantisation arithmetic is the oracle's (oracle/ep.py quant_int8_rows, eps = 1e-12)."""
import os
import time

import torch
import torch.distributed as dist


def _quant_rows(x_bf16: torch.Tensor):
    """Per-token dynamic INT8 (cam_moe_dispatch_normal.h:326-363 arithmetic): s = 127 / (amax + 1e-12), q = rint(x * s), scale = 1 / s."""
    xf = x_bf16.float()
    amax = xf.abs().amax(dim=1)
    # tensor / tensor: a scalar numerator makes ATen use a vectorised reciprocal that is not correctly rounded
    s = torch.full_like(amax, 127.0) / (amax + 1e-12)
    q = torch.round(xf * s[:, None]).to(torch.int8)
    return q, torch.ones_like(s) / s


def layout(topk_idx: torch.Tensor, num_experts: int, group):
    W, rank = dist.get_world_size(group), dist.get_rank(group)
    L = num_experts // W
    valid = topk_idx.reshape(-1)
    local_cnt = torch.bincount(valid[valid >= 0], minlength=num_experts)                    # histc :497-499
    input_splits = local_cnt.reshape(W, L).sum(dim=1).tolist()
    gathered = [torch.empty_like(local_cnt) for _ in range(W)]
    dist.all_gather(gathered, local_cnt, group=group)                                       # :509-511
    glob = torch.stack(gathered)                                                            # [W, E]
    mine = glob[:, rank * L:(rank + 1) * L]                                                 # num_global_tokens_per_local_expert
    output_splits = mine.sum(dim=1).tolist()
    ids = torch.arange(L).repeat(W)
    global_tokens_indices = torch.repeat_interleave(ids, mine.reshape(-1))                  # :538-545
    return dict(L=L, input_splits=input_splits, output_splits=output_splits, per_expert=mine.sum(dim=0),
                global_tokens_indices=global_tokens_indices)


def dispatch(x_bf16, topk_idx, lay, group, quant=True):
    T, K = topk_idx.shape
    flat = topk_idx.reshape(-1)
    order = torch.argsort(torch.where(flat >= 0, flat, torch.iinfo(flat.dtype).max), stable=True)   # init_routing: expert-sorted, stable
    n_valid = int((flat >= 0).sum())
    order = order[:n_valid]
    src_row = order // K
    if quant:
        q, sc = _quant_rows(x_bf16)
        send, send_scale = q.index_select(0, src_row), sc.index_select(0, src_row)
    else:
        send, send_scale = x_bf16.index_select(0, src_row), None
    n_recv = sum(lay["output_splits"])
    recv = torch.empty((n_recv,) + tuple(send.shape[1:]), dtype=send.dtype)
    dist.all_to_all_single(recv, send, lay["output_splits"], lay["input_splits"], group=group)          # :653-660
    recv_scale = None
    if quant:
        recv_scale = torch.empty(n_recv, dtype=torch.float32)
        dist.all_to_all_single(recv_scale, send_scale, lay["output_splits"], lay["input_splits"], group=group)   # :646-651
    if lay["L"] > 1:                                                                        # second routing pass :662-698
        order2 = torch.argsort(lay["global_tokens_indices"], stable=True)
        recv = recv.index_select(0, order2)
        if quant:
            recv_scale = recv_scale.index_select(0, order2)
    else:
        order2 = None
    handle = dict(order=order, order2=order2, T=T, K=K, input_splits=lay["input_splits"], output_splits=lay["output_splits"])
    return recv, recv_scale, handle


def combine(y_bf16, handle, topk_weights, group):
    if handle["order2"] is not None:                                                        # finalize_routing of the global sort :751-761
        unsorted = torch.empty_like(y_bf16)
        unsorted[handle["order2"]] = y_bf16
        y_bf16 = unsorted
    back = torch.empty((sum(handle["input_splits"]), y_bf16.shape[1]), dtype=y_bf16.dtype)
    dist.all_to_all_single(back, y_bf16, handle["input_splits"], handle["output_splits"], group=group)  # :763-770
    T, K = handle["T"], handle["K"]
    order = handle["order"]
    w = topk_weights.reshape(-1).index_select(0, order).float()
    out = torch.zeros((T, y_bf16.shape[1]), dtype=torch.float32)
    out.index_add_(0, order // K, back.float() * w[:, None])                               # finalize_routing with scales :772-781
    return out.to(torch.bfloat16)


def one_pass(x, topk_idx, topk_w, num_experts, group, quant=True):
    lay = layout(topk_idx, num_experts, group)
    recv, scale, handle = dispatch(x, topk_idx, lay, group, quant)
    y = (recv.float() * scale[:, None]).to(torch.bfloat16) if quant else recv               # expert stand-in: per_token_cast_back
    out = combine(y, handle, topk_w, group)
    return out, recv, scale, lay


def _worker(rank, W, port, T, H, K, E, threads, min_seconds, max_passes, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(max(1, threads))
    dist.init_process_group("gloo", rank=rank, world_size=W)
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn((T, H), generator=g).to(torch.bfloat16)
    topk_idx = torch.topk(torch.randn((T, E), generator=g).abs() + 1, K, dim=-1, sorted=False)[1]
    topk_w = torch.randn((T, K), generator=g)
    one_pass(x, topk_idx, topk_w, E, dist.group.WORLD)                                      # warm-up (connections, allocator)
    dist.barrier()
    t0 = time.perf_counter()
    passes, rows = 0, 0
    while True:
        _, recv, _, _ = one_pass(x, topk_idx, topk_w, E, dist.group.WORLD)
        passes += 1
        rows = recv.shape[0]
        stop = torch.tensor([1.0 if (time.perf_counter() - t0 > min_seconds or passes >= max_passes) else 0.0])
        dist.all_reduce(stop, op=dist.ReduceOp.MAX)                                         # every rank stops after the same pass
        if stop.item() > 0:
            break
    dist.barrier()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(rows)])
    dist.all_reduce(tot)
    if rank == 0:
        ret["seconds"], ret["passes"], ret["rows_all_ranks"] = dt, passes, float(tot.item())
    dist.destroy_process_group()


def timed_run(W=8, T=4096, H=7168, K=8, E=256, cores=None, min_seconds=10.0, max_passes=16, port=29655):
    """Spawn W gloo ranks sharing `cores` host threads and time whole passes of layout + dispatch(int8) + combine.
    -> dict(seconds, passes, rows_all_ranks, cores, threads_per_rank)."""
    import torch.multiprocessing as mp

    cores = cores or os.cpu_count() or W
    threads = max(1, cores // W)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(W, port, T, H, K, E, threads, min_seconds, max_passes, ret), nprocs=W, join=True)
    out = dict(ret)
    out["cores"], out["threads_per_rank"] = threads * W, threads
    return out
