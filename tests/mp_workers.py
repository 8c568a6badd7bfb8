"""Worker functions for the multi-process tests (torch.multiprocessing.spawn needs importable top-level functions)."""
import os
import sys
import traceback

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "sgl-kernel-npu_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)


def make_topk(rng, T, K, E, drop=0.0):
    scores = np.abs(rng.standard_normal((T, E))) + 1
    idx = np.argsort(-scores, axis=1, kind="stable")[:, :K].astype(np.int64)
    if drop > 0:
        idx[rng.random((T, K)) < drop] = -1
    return idx


def make_inputs(W, T, H, K, E, drop, seed=1234):
    from oracle.bf16 import f32_to_bf16_bits_rne
    rng = np.random.default_rng(seed)
    Ts = [T + r for r in range(W)]
    xs = [f32_to_bf16_bits_rne((rng.standard_normal((t, H)) * 2).astype(np.float32)) for t in Ts]
    idxs = [make_topk(rng, t, K, E, drop) for t in Ts]
    ws = [rng.standard_normal((t, K)).astype(np.float32) for t in Ts]
    return xs, idxs, ws


def _set_device(rank):
    """The GPU a worker process runs on.  Default: cuda:0 for every rank (W processes share one GPU, windows mapped through hipIpc -- what the
    one-GPU test boxes offer).  MI_TEST_DEVICE_PER_RANK=1 on a multi-GPU node: rank r takes GPU r mod device_count, so the same parity tests run with
    every rank's window behind its own GPU and the peers' stores crossing xGMI -- the first thing to run when a node is available
    (`MI_TEST_DEVICE_PER_RANK=1 python -m pytest tests/test_deep_ep_gpu.py -m gpu -x -q`)."""
    n = torch.cuda.device_count()
    dev = rank % n if (os.environ.get("MI_TEST_DEVICE_PER_RANK") == "1" and n > 1) else 0
    torch.cuda.set_device(dev)
    return dev


def _init(rank, world, port, backend="gloo"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist.group.WORLD


def run_guarded(fn, rank, *args):
    try:
        fn(rank, *args)
    except Exception:
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(17)


# ----------------------------------------------------------------------------------------------
# CPU / gloo: host plumbing of deep_ep.Buffer + alltoall strategies with the oracle-backed test double
# ----------------------------------------------------------------------------------------------
def cpu_alltoall_worker(rank, world, port, cfg):
    run_guarded(_cpu_alltoall, rank, world, port, cfg)


def _cpu_alltoall(rank, world, port, cfg):
    import deep_ep
    from fake_runtime import FakeRuntime
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits
    group = _init(rank, world, port)
    W, T, H, K, E, drop, quant = cfg
    assert W == world
    deep_ep.Buffer._runtime_factory = staticmethod(lambda *a: FakeRuntime(*a))
    buf = deep_ep.Buffer(group, normal_strategy="alltoall", low_latency_strategy="alltoall", low_latency_mode=True)
    assert buf.normal_strategy.get_name() == "alltoall"
    xs, idxs, ws = make_inputs(W, T, H, K, E, drop)
    x, ti, tw = bits_to_torch(xs[rank]), torch.from_numpy(idxs[rank]), torch.from_numpy(ws[rank])
    per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
    want = O.normal_dispatch(xs, idxs, E, quant)
    assert np.array_equal(per_expert.numpy(), want[rank].layout["num_tokens_per_expert"])
    recv_x, _, _, lst, handle, ev = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                 num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                 quant_mode="int8" if quant else None)
    ev.current_stream_wait()
    n = want[rank].total_recv
    assert lst == want[rank].num_recv_tokens_per_expert_list
    assert len(handle) == 8 and handle[7] is tw and handle[6] is ti
    assert np.array_equal(handle[3].numpy()[:3 * n], want[rank].recv_src_idx[:3 * n])
    assert np.array_equal(handle[5].numpy(), want[rank].send_head)
    if quant:
        assert np.array_equal(recv_x[0].numpy()[:n], want[rank].recv_x[:n])
        assert np.array_equal(recv_x[1].numpy()[:n], want[rank].recv_x_scales[:n])
        y = bits_to_torch(O.per_token_cast_back(want[rank].recv_x, want[rank].recv_x_scales))
    else:
        assert np.array_equal(torch_to_bits(recv_x)[:n], want[rank].recv_x[:n])
        y = recv_x
    ys = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
    comb_want = O.combine(ys, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
    out, _, _ = buf.combine(y, handle)
    assert np.array_equal(torch_to_bits(out), comb_want[rank])
    # low-latency pair through the same transport
    MT = T + W
    llw = O.low_latency_dispatch(xs, idxs, MT, E, quant)
    rx, cnt, h, _, hook = buf.low_latency_dispatch(x, ti, MT, E, use_fp8=quant)
    hook()
    assert np.array_equal(cnt.numpy(), llw[rank].packed_recv_count)
    assert np.array_equal(h[1].numpy(), llw[rank].layout_range)
    nn = llw[rank].total
    if quant:
        assert np.array_equal(rx[0].numpy()[:nn], llw[rank].packed_recv_x[:nn])
    else:
        assert np.array_equal(torch_to_bits(rx)[:nn], llw[rank].packed_recv_x[:nn])
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: W processes sharing ONE GPU, real deep_ep_cpp runtime, hipIpc-mapped windows
# ----------------------------------------------------------------------------------------------
def gpu_buffer_worker(rank, world, port, cfg):
    run_guarded(_gpu_buffer, rank, world, port, cfg)


def _gpu_buffer(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits
    _set_device(rank)
    W, T, H, K, E, drop, quant, strategy, iters = cfg
    # RCCL refuses several ranks on one GPU, so the multi-process cases bootstrap over gloo; the alltoall strategy
    # needs a real RCCL group for its device collectives and is exercised at world size 1 here
    group = _init(rank, world, port, "nccl" if strategy == "alltoall" else "gloo")
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(768 << 20))
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "20000")
    buf = deep_ep.Buffer(group, low_latency_mode=True, normal_strategy=strategy, low_latency_strategy=strategy)
    assert buf.normal_strategy.get_name() == strategy, (buf.normal_strategy.get_name(), buf.p2p_available)
    for it in range(iters):
        # the first of several iterations is small, so the next one receives far more rows than the runtime's speculative
        # receive buffers (sized from the previous call) hold: exercises both the hit and the re-pull path
        Tt = max(8, T // 4) if (iters > 1 and it == 0) else T
        if strategy == "default":
            # both normal-dispatch transports, alternating (every rank switches at the same call): identical results required
            buf.runtime.set_dispatch_transport(("push", "pull")[it % 2])
            assert buf.runtime.get_dispatch_transport() == ("push", "pull")[it % 2]
        xs, idxs, ws = make_inputs(W, Tt, H, K, E, drop, seed=100 + it)
        x = bits_to_torch(xs[rank]).cuda()
        ti = torch.from_numpy(idxs[rank]).cuda()
        tw = torch.from_numpy(ws[rank]).cuda()
        want = O.normal_dispatch(xs, idxs, E, quant)
        per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
        # diagnose tensors (reference buffer.py:343-345,500-501): accumulated microseconds per rank, filled when passed
        wait_stats = torch.zeros(W, dtype=torch.int32, device="cuda") if it == iters - 1 else None
        send_stats = torch.zeros(W, dtype=torch.int32, device="cuda") if it == iters - 1 else None
        recv_x, _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                    num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                    quant_mode="int8" if quant else None,
                                                    dispatch_wait_recv_cost_stats=wait_stats)
        n = want[rank].total_recv
        assert lst == want[rank].num_recv_tokens_per_expert_list, (lst, want[rank].num_recv_tokens_per_expert_list)
        assert np.array_equal(handle[3].cpu().numpy()[:3 * n], want[rank].recv_src_idx[:3 * n])
        assert np.array_equal(handle[5].cpu().numpy(), want[rank].send_head)
        if quant:
            assert np.array_equal(recv_x[0].cpu().numpy()[:n], want[rank].recv_x[:n])
            assert np.array_equal(recv_x[1].cpu().numpy()[:n], want[rank].recv_x_scales[:n])
            y = bits_to_torch(O.per_token_cast_back(want[rank].recv_x, want[rank].recv_x_scales)).cuda()
        else:
            assert np.array_equal(torch_to_bits(recv_x)[:n], want[rank].recv_x[:n])
            y = recv_x
        ys = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
        comb_want = O.combine(ys, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
        out, _, _ = buf.combine(y, handle, combine_send_cost_stats=send_stats)
        assert np.array_equal(torch_to_bits(out), comb_want[rank]), "combine mismatch"
        if wait_stats is not None and strategy == "default":
            ws_, ss_ = wait_stats.cpu(), send_stats.cpu()
            assert (ws_ >= 0).all() and (ws_ < 60_000_000).all() and (ss_ >= 0).all() and (ss_ < 60_000_000).all()
            assert int(ss_.min()) == int(ss_.max())             # every destination is charged the push duration
        if strategy == "default" and it == iters - 1:
            # DeepEP graph mode (buffer.py:337-338,356-358): worst-case sized outputs, no host sync, empty count list
            worst = Tt * K * W
            rw, _, _, lst_w, handle_w, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                        num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                        quant_mode="int8" if quant else None, num_worst_tokens=worst)
            assert lst_w == []
            rxw = rw[0] if quant else rw
            assert rxw.shape[0] == worst
            if quant:
                assert np.array_equal(rxw.cpu().numpy()[:n], want[rank].recv_x[:n])
            else:
                assert np.array_equal(torch_to_bits(rxw)[:n], want[rank].recv_x[:n])
            assert np.array_equal(handle_w[3].cpu().numpy()[:3 * n], want[rank].recv_src_idx[:3 * n])
            out_w, _, _ = buf.combine(y, handle_w)
            assert np.array_equal(torch_to_bits(out_w), comb_want[rank]), "combine after worst-token dispatch mismatch"
        # low latency
        MT = T + W
        llw = O.low_latency_dispatch(xs, idxs, MT, E, quant)
        rx, cnt, h, _, hook = buf.low_latency_dispatch(x, ti, MT, E, use_fp8=quant)
        hook()
        assert np.array_equal(cnt.cpu().numpy(), llw[rank].packed_recv_count)
        assert np.array_equal(h[1].cpu().numpy(), llw[rank].layout_range)
        nn = llw[rank].total
        assert np.array_equal(h[0].cpu().numpy()[:3 * nn], llw[rank].src_info)
        if quant:
            assert np.array_equal(rx[0].cpu().numpy()[:nn], llw[rank].packed_recv_x[:nn])
            assert np.array_equal(rx[1].cpu().numpy()[:nn], llw[rank].packed_recv_x_scales[:nn])
            yl = bits_to_torch(O.per_token_cast_back(llw[rank].packed_recv_x, llw[rank].packed_recv_x_scales)).cuda()
        else:
            assert np.array_equal(torch_to_bits(rx)[:nn], llw[rank].packed_recv_x[:nn])
            yl = rx
        yls = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) if quant else w.packed_recv_x for w in llw]
        wabs = [np.abs(w_) for w_ in ws]
        llc_want = O.combine(yls, [w.src_info for w in llw], [w.total for w in llw], idxs, wabs, E)
        outl, _, hook = buf.low_latency_combine(yl, ti, torch.from_numpy(wabs[rank]).cuda(), h)
        hook()
        assert np.array_equal(torch_to_bits(outl), llc_want[rank]), "LL combine mismatch"
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: low-latency pair with a rank that has NO tokens (and ragged batches on the others), every launch form
# ----------------------------------------------------------------------------------------------
def gpu_ll_empty_rank_worker(rank, world, port, cfg):
    run_guarded(_gpu_ll_empty_rank, rank, world, port, cfg)


def _gpu_ll_empty_rank(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits
    _set_device(rank)
    W, T0, H, K, E, quant, forms = cfg[:7]
    # optional 8th entry: stale_rank (that rank fails the in-launch self-test leg for everybody: DEEPEP_SELF_TEST_STALE_RANK), own_gpu (pretend
    # every rank owns its GPU: default forms, waiting launches uncapped), tokens (per-rank token counts) + max_tokens, repeat_combine
    extra = cfg[7] if len(cfg) > 7 else {}
    if forms is not None:
        os.environ["MI_EP_LL_FUSED_COUNTS"], os.environ["MI_EP_COMBINE_FUSED"] = forms
    else:
        os.environ.pop("MI_EP_LL_FUSED_COUNTS", None), os.environ.pop("MI_EP_COMBINE_FUSED", None)
    if "stale_rank" in extra:
        os.environ["DEEPEP_SELF_TEST_STALE_RANK"] = str(extra["stale_rank"])
    group = _init(rank, world, port, "gloo")
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(256 << 20))
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "20000")
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        buf = deep_ep.Buffer(group, low_latency_mode=True)
    if extra.get("own_gpu"):
        buf.runtime.set_ranks_share_device(False)
    if "stale_rank" in extra and W > 1:
        # the windows passed their first leg, the in-launch hand-off did not: the low-latency calls keep the window strategies on their
        # three-launch forms, whatever the env asks for, and the Buffer said so
        assert buf.p2p_available and not buf.runtime.get_two_launch_forms()
        assert list(buf.runtime.get_low_latency_launch_forms()) == [0, 0]
        assert any("three-launch forms" in str(c.message) for c in caught), [str(c.message) for c in caught]
    elif W > 1:
        assert buf.runtime.get_two_launch_forms()
        if extra.get("own_gpu") and forms is None:
            assert list(buf.runtime.get_low_latency_launch_forms()) == [2, 2]
    if "tokens" in extra:
        from oracle.bf16 import f32_to_bf16_bits_rne
        rng = np.random.default_rng(4321)
        xs = [f32_to_bf16_bits_rne((rng.standard_normal((t, H)) * 2).astype(np.float32)) for t in extra["tokens"]]
        idxs = [make_topk(rng, t, K, E, 0.2) for t in extra["tokens"]]
        ws = [rng.standard_normal((t, K)).astype(np.float32) for t in extra["tokens"]]
        MT = extra["max_tokens"]
    else:
        xs, idxs, ws = make_inputs(W, T0, H, K, E, 0.2)          # rank r has T0 + r tokens: T0 = 0 leaves rank 0 without any
        MT = T0 + W
    x, ti = bits_to_torch(xs[rank]).cuda(), torch.from_numpy(idxs[rank]).cuda()
    wabs = [np.abs(w_) for w_ in ws]
    llw = O.low_latency_dispatch(xs, idxs, MT, E, quant)
    yls = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) if quant else w.packed_recv_x for w in llw]
    llc_want = O.combine(yls, [w.src_info for w in llw], [w.total for w in llw], idxs, wabs, E)
    for it in range(3):          # three calls: both ping-pong halves and a reused one
        rx, cnt, h, _, hook = buf.low_latency_dispatch(x, ti, MT, E, use_fp8=quant)
        hook()
        assert np.array_equal(cnt.cpu().numpy(), llw[rank].packed_recv_count), it
        assert np.array_equal(h[1].cpu().numpy(), llw[rank].layout_range), it
        nn = llw[rank].total
        assert np.array_equal(h[0].cpu().numpy()[:3 * nn], llw[rank].src_info), it
        if quant:
            assert np.array_equal(rx[0].cpu().numpy()[:nn], llw[rank].packed_recv_x[:nn]), it
            assert np.array_equal(rx[1].cpu().numpy()[:nn], llw[rank].packed_recv_x_scales[:nn]), it
            yl = bits_to_torch(O.per_token_cast_back(llw[rank].packed_recv_x, llw[rank].packed_recv_x_scales)).cuda()
        else:
            assert np.array_equal(torch_to_bits(rx)[:nn], llw[rank].packed_recv_x[:nn]), it
            yl = rx
        outl, _, hook = buf.low_latency_combine(yl, ti, torch.from_numpy(wabs[rank]).cuda(), h)
        hook()
        assert tuple(outl.shape) == (xs[rank].shape[0], H)
        assert np.array_equal(torch_to_bits(outl), llc_want[rank]), (it, "LL combine mismatch")
        if extra.get("repeat_combine"):
            # a second combine on the same handle, no dispatch in between: it takes the three-launch form on every rank (a two-launch
            # combine is only safe behind a dispatch's all-to-all count exchange) and returns the same sums
            if W > 1 and forms is not None and forms[1] == "2":
                assert list(buf.runtime.get_low_latency_launch_forms())[1] == 0
            for rep in range(2):
                outl2, _, hook = buf.low_latency_combine(yl, ti, torch.from_numpy(wabs[rank]).cuda(), h)
                hook()
                assert np.array_equal(torch_to_bits(outl2), llc_want[rank]), (it, rep, "repeated LL combine mismatch")
    if extra.get("capture_lone_combine") and xs[rank].shape[0] > 0:
        # a combine recorded into a graph WITHOUT its dispatch is replayed combine after combine: it must have been captured in the
        # three-launch form (whose signal / wait is all-to-all), and every replay returns the same sums
        rx, cnt, h, _, hook = buf.low_latency_dispatch(x, ti, MT, E, use_fp8=quant)
        hook()
        wt = torch.from_numpy(wabs[rank]).cuda()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buf.low_latency_combine(yl, ti, wt, h)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            outg, _, _ = buf.low_latency_combine(yl, ti, wt, h)
        for rep in range(3):
            g.replay()
            torch.cuda.synchronize()
            assert np.array_equal(torch_to_bits(outg), llc_want[rank]), (rep, "replayed lone combine mismatch")
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: BASELINE C2 at full size through deep_ep.Buffer: 8 processes (one GPU here, 8 GPUs in production), 4096 tokens per rank
# ----------------------------------------------------------------------------------------------
def gpu_c2_size_worker(rank, world, port, cfg):
    run_guarded(_gpu_c2_size, rank, world, port, cfg)


def _c2_inputs(s, T, H, K, E):
    g = torch.Generator(device="cuda").manual_seed(4321 + s)
    x = torch.randn((T, H), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    idx = torch.topk(torch.randn((T, E), generator=g, device="cuda").abs() + 1, K, dim=-1, sorted=False)[1]
    w = torch.rand((T, K), generator=g, device="cuda") + 0.5
    return x, idx, w


def _gpu_c2_size(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import torch_to_bits
    _set_device(rank)
    import faulthandler
    import time
    W, T, H, K, E = cfg
    L = E // W
    faulthandler.dump_traceback_later(240, exit=True)      # a stuck rank prints where it is and ends the test instead of hanging it
    t_start = time.time()
    trace = (lambda *a: print(f"[c2 rank {rank} +{time.time() - t_start:6.1f}s]", *a, file=sys.stderr, flush=True)) \
        if os.getenv("MI_TEST_TRACE") else (lambda *a: None)
    group = _init(rank, world, port)
    os.environ["DEEPEP_WINDOW_BYTES"] = str(6 * (T * K * H * 2 + (8 << 20)) + (4 << 20))     # combine slots are the largest region
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "60000")
    buf = deep_ep.Buffer(group)
    trace("buffer ready")
    x, ti, tw = _c2_inputs(rank, T, H, K, E)
    hist = torch.zeros(E, dtype=torch.long, device="cuda")
    idx_all = []
    for s in range(W):
        i_s = _c2_inputs(s, T, H, K, E)[1]
        idx_all.append(i_s)
        hist += torch.bincount(i_s.reshape(-1), minlength=E)
    first = None
    for transport in ("push", "pull", "push"):
        buf.runtime.set_dispatch_transport(transport)
        trace("dispatch", transport)
        per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
        (rx, rs), _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                      num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                      quant_mode="int8")
        trace("dispatched", sum(lst))
        assert lst == hist[rank * L:(rank + 1) * L].tolist()                 # counts == global histogram (test_intranode.py:401-411)
        n = sum(lst)
        assert rx.shape[0] == max(n, 1)
        tri = handle[3][:3 * n].view(-1, 3).long()
        e_row = torch.empty(n, dtype=torch.long, device="cuda")
        for s in range(W):
            m = tri[:, 0] == s
            e_row[m] = idx_all[s][tri[m, 1], tri[m, 2]]
        key = ((e_row - rank * L) * W + tri[:, 0]) * (T * K) + tri[:, 1] * K + tri[:, 2]
        assert bool((key[1:] > key[:-1]).all()), "receive order is not (local expert, source rank, row-major (t, k))"
        assert bool(((e_row >= rank * L) & (e_row < (rank + 1) * L)).all())
        # sampled rows: INT8 payload + scale == the oracle's quantisation of the source row
        sel = torch.randperm(n, device="cuda")[:1024]
        src_rows = torch.empty((sel.numel(), H), dtype=torch.bfloat16, device="cuda")
        for s in range(W):
            m = tri[sel, 0] == s
            if bool(m.any()):
                src_rows[m] = _c2_inputs(s, T, H, K, E)[0][tri[sel][m, 1]]
        q_want, s_want = O.quant_int8_rows(torch_to_bits(src_rows), 1e-12)
        assert np.array_equal(rx[sel].cpu().numpy(), q_want), transport
        assert np.array_equal(rs[sel].cpu().numpy().view(np.uint32), s_want.view(np.uint32)), transport
        if first is None:
            first = (rx.clone(), rs.clone(), handle[3].clone())
        else:                                                                # transports and repeated calls: identical bytes
            assert torch.equal(first[0], rx) and torch.equal(first[1], rs) and torch.equal(first[2][:3 * n], handle[3][:3 * n])
        # round trip (test_intranode.py:431-444): combine(dequantised rows) == x * sum_k w
        y = (rx.float() * rs[:, None]).to(torch.bfloat16)
        trace("combine")
        out, _, _ = buf.combine(y, handle)
        torch.cuda.synchronize()
        trace("combined")
        golden = x.float() * tw.sum(dim=1, keepdim=True)
        a, b = out.double() + 1, golden.double() + 1
        diff = 1 - 2 * (a * b).sum() / (a * a + b * b).sum()
        assert diff.item() < 3e-3, diff.item()
    torch.cuda.synchronize()
    faulthandler.cancel_dump_traceback_later()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: fused_deep_moe through deep_ep.Buffer vs the oracle
# ----------------------------------------------------------------------------------------------
def gpu_fused_moe_worker(rank, world, port, cfg):
    run_guarded(_gpu_fused_moe, rank, world, port, cfg)


def _gpu_fused_moe(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits, bf16_bits_to_f32
    _set_device(rank)
    W, T, H, I, K, E, layout = cfg
    group = _init(rank, world, port)
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(1 << 30))
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    L = E // W
    rng = np.random.default_rng(7)
    xs, idxs, _ = make_inputs(W, T, H, K, E, 0.1, seed=5)
    xs = [x[:T] for x in xs]
    idxs = [i[:T] for i in idxs]
    ws = [np.abs(rng.standard_normal((T, K))).astype(np.float32) for _ in range(W)]
    # weights as in tests/python/deepep/test_fused_deep_moe.py:32-44 (randint(-16,16), scales U*4e-4+1.5e-3), per rank
    w13 = [rng.integers(-16, 16, (L, 2 * I, H)).astype(np.int8) for _ in range(W)]
    w2 = [rng.integers(-16, 16, (L, H, I)).astype(np.int8) for _ in range(W)]
    s13 = [(rng.random((L, 2 * I)) * 4e-4 + 1.5e-3).astype(np.float32) for _ in range(W)]
    s2 = [(rng.random((L, H)) * 4e-4 + 1.5e-3).astype(np.float32) for _ in range(W)]
    want = O.fused_deep_moe(xs, idxs, ws, w13, s13, w2, s2, T, E)[rank]
    perm = O.permute_fusion_cols(2 * I)
    w13_p = torch.from_numpy(np.ascontiguousarray(w13[rank][:, perm, :])).cuda()          # [L, 2I, H], fusion-tile order
    s13_p = torch.from_numpy(np.ascontiguousarray(s13[rank][:, perm])).cuda()
    w2_t = torch.from_numpy(w2[rank]).cuda()
    if layout == "reference":        # logical shapes of the reference: [L, H, 2I] and [L, I, H]
        w13_p = w13_p.transpose(1, 2).contiguous()
        w2_t = w2_t.transpose(1, 2).contiguous()
    x = bits_to_torch(xs[rank]).cuda()
    ti = torch.from_numpy(idxs[rank]).cuda()
    tw = torch.from_numpy(ws[rank]).cuda()
    ll = O.low_latency_dispatch(xs, idxs, T, E, True)[rank]
    if layout == "ffn":
        # FuseMode.DISPATCH_FFN_COMBINE (tests/python/deepep/test_dispatch_ffn_combine.py:72-88,390-405): plain [L, H, 2I] /
        # [L, I, H] weights (gate = first I columns), fp32 scale bits widened to int64, max_output_size = T*K*W
        as_i64 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32).astype(np.int64)).cuda()
        w13_f2 = torch.from_numpy(np.ascontiguousarray(w13[rank].transpose(0, 2, 1))).cuda()
        w2_f2 = torch.from_numpy(np.ascontiguousarray(w2[rank].transpose(0, 2, 1))).cuda()
        for _ in range(2):
            out, nums = buf.fused_deep_moe(x, ti, tw, w13_f2, as_i64(s13[rank]), w2_f2, as_i64(s2[rank]), T * K * W, E, 1, 2)
        per_expert = np.diff(np.concatenate([[0], ll.layout_range.reshape(L, W)[:, -1]]))
        assert nums.shape == (L,) and np.array_equal(nums.cpu().numpy(), per_expert)   # test_dispatch_ffn_combine.py:425-440
    else:
        for _ in range(2):
            out, ep_recv_count = buf.fused_deep_moe(x, ti, tw, w13_p, s13_p, w2_t, torch.from_numpy(s2[rank]).cuda(), T, E)
        assert np.array_equal(ep_recv_count.cpu().numpy(), ll.layout_range)           # recv counts exact (test :519-521)
    assert out.shape == (T, H) and out.dtype == torch.bfloat16
    got = bf16_bits_to_f32(torch_to_bits(out))
    ref = bf16_bits_to_f32(want)
    diff = O.calc_diff(got, ref)
    denom = np.maximum(np.abs(ref), 1e-2)
    assert diff < 1e-5, diff
    assert np.mean(np.abs(got - ref) / denom) < 4e-4, np.mean(np.abs(got - ref) / denom)   # reference: avg_diff < 4e-4 (:470)
    # prefill-size batches multiply the staged token rows in place (row-offset table, mi_ep_dispatch_resolve_rows +
    # mi_ep_moe_gemm1_swiglu_rows); with the K-fold gathered copy instead the output must be the same BITS
    if layout != "ffn" and hasattr(buf.runtime, "set_fused_rows_in_place"):
        assert buf.runtime.get_fused_rows_in_place()
        buf.runtime.set_fused_rows_in_place(False)
        out2, cnt2 = buf.fused_deep_moe(x, ti, tw, w13_p, s13_p, w2_t, torch.from_numpy(s2[rank]).cuda(), T, E)
        buf.runtime.set_fused_rows_in_place(True)
        out3, _ = buf.fused_deep_moe(x, ti, tw, w13_p, s13_p, w2_t, torch.from_numpy(s2[rank]).cuda(), T, E)
        assert torch.equal(out2, out) and torch.equal(out3, out) and torch.equal(cnt2, ep_recv_count)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: BASELINE C5 at full size (EP = 8, 4096 tok/rank, hidden 7168, 2I = 4096, 32 local experts per rank) through
# deep_ep.Buffer.fused_deep_moe: the prefill-size exchange branch with peers; a sample of every rank's tokens against the
# per-token float64 evaluation (tests/fused_f64.py) at the reference bar (test_fused_deep_moe.py:470)
# ----------------------------------------------------------------------------------------------
def gpu_fused_c5_worker(rank, world, port, cfg):
    run_guarded(_gpu_fused_c5, rank, world, port, cfg)


def _gpu_fused_c5(rank, world, port, cfg):
    import faulthandler
    import deep_ep
    import fused_f64 as F
    _set_device(rank)
    W, T, H, I, K, L, samples = cfg
    E = L * W
    faulthandler.dump_traceback_later(420, exit=True)
    group = _init(rank, world, port)
    cb = T * K * H * 2
    os.environ["DEEPEP_WINDOW_BYTES"] = str(min(6 << 30, 6 * (cb + (8 << 20)) + (4 << 20)))
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "120000")
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    w13, w2, s13, s2 = F.fused_weights(900 + rank, L, H, I)
    perm = F.fusion_perm(2 * I)
    w13_p, s13_p = w13[:, perm, :].contiguous(), s13[:, perm].contiguous()
    del w13
    g = torch.Generator(device="cuda").manual_seed(1900 + rank)
    x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
    idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
    idx[5, 1] = -1
    w = torch.rand((T, K), generator=g, device="cuda")
    with torch.inference_mode():                      # the reference harness runs its ranks under inference_mode (no version counters)
        for _ in range(2):
            out, counts = buf.fused_deep_moe(x, idx, w, w13_p, s13_p, w2, s2, T, E)
    torch.cuda.synchronize()
    assert out.shape == (T, H) and out.dtype == torch.bfloat16 and bool(torch.isfinite(out.float()).all())
    # received rows per (local expert, source rank): inclusive cumulative counts == the global histogram
    hist = torch.zeros((W, E), dtype=torch.long, device="cuda")
    for s in range(W):
        gs = torch.Generator(device="cuda").manual_seed(1900 + s)
        torch.randn((T, H), generator=gs, device="cuda")
        i_s = torch.topk(torch.rand((T, E), generator=gs, device="cuda"), K, dim=-1)[1]
        i_s[5, 1] = -1
        hist[s] = torch.bincount(i_s[i_s >= 0].reshape(-1), minlength=E)
    mine = hist[:, rank * L:(rank + 1) * L].t().reshape(-1)                    # [L, W] -> (le, src)
    assert torch.equal(counts.long().reshape(-1), torch.cumsum(mine, 0)), "recv counts differ from the global histogram"
    dist.barrier()                                    # every rank done with the fused calls before the memory-hungry check
    del w13_p, s13_p, w2, s2
    torch.cuda.empty_cache()
    r = F.sampled_check(out, x, idx, w, lambda rr: F.fused_weights(900 + rr, L, H, I), L, n_samples=samples, seed=rank)
    assert r["ok"], r
    faulthandler.cancel_dump_traceback_later()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: the whole API under torch.inference_mode() (the reference harness decorates its rank function with it,
# tests/python/deepep/test_fused_deep_moe_a5.py:723): inference tensors carry no version counter, every call must still work
# and give the bytes the same call gives outside inference mode
# ----------------------------------------------------------------------------------------------
def gpu_inference_mode_worker(rank, world, port, cfg):
    run_guarded(_gpu_inference_mode, rank, world, port, cfg)


def _gpu_inference_mode(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch
    _set_device(rank)
    W, T, H, I, K, E = cfg
    L = E // W
    group = _init(rank, world, port)
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(512 << 20))
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    xs, idxs, ws = make_inputs(W, T, H, K, E, 0.1, seed=11)
    rng = np.random.default_rng(3)
    w13 = rng.integers(-16, 16, (L, 2 * I, H)).astype(np.int8)
    w2 = rng.integers(-16, 16, (L, I, H)).astype(np.int8)          # reference logical shape [L, I, H]: goes through the weight cache
    s13 = (rng.random((L, 2 * I)) * 4e-4 + 1.5e-3).astype(np.float32)
    s2 = (rng.random((L, H)) * 4e-4 + 1.5e-3).astype(np.float32)

    def run():
        x = bits_to_torch(xs[rank]).cuda()
        ti = torch.from_numpy(idxs[rank]).cuda()
        tw = torch.from_numpy(np.abs(ws[rank])).cuda()
        res = []
        for _ in range(2):      # twice: the second call meets the layout stash / weight cache entries the first one left
            per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
            (rx, rs), _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                          num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                          quant_mode="int8")
            y = (rx.float() * rs[:, None]).to(torch.bfloat16)
            out, _, _ = buf.combine(y, handle)
            (lx, ls), cnt, h, _, hook = buf.low_latency_dispatch(x, ti, T + W, E, use_fp8=True)      # the same bound on every rank
            hook()
            n = int(h[1].reshape(-1)[-1])
            yl = (lx.float() * ls[:, None]).to(torch.bfloat16)
            outl, _, hook = buf.low_latency_combine(yl, ti, tw, h)
            hook()
            fo, _ = buf.fused_deep_moe(x, ti, tw, torch.from_numpy(w13).cuda(), torch.from_numpy(s13).cuda(),
                                       torch.from_numpy(w2).cuda(), torch.from_numpy(s2).cuda(), T + W, E)
            # rewriting the routing in place between calls must be seen (inference tensors: the stash is never reused)
            ti = ti.clone()
            res.append([t.clone() for t in (rx[:sum(lst)], rs[:sum(lst)], out, lx[:n], outl, fo)])
        return res

    plain = run()
    with torch.inference_mode():
        inf = run()
        assert inf[0][2].is_inference()
    # and the low-latency combine of both against the oracle (same inputs every iteration)
    from oracle.bf16 import torch_to_bits
    MT = T + W
    llw = O.low_latency_dispatch(xs, idxs, MT, E, True)
    yls = [O.per_token_cast_back(w_.packed_recv_x, w_.packed_recv_x_scales) for w_ in llw]
    llc = O.combine(yls, [w_.src_info for w_ in llw], [w_.total for w_ in llw], idxs, [np.abs(w_) for w_ in ws], E)[rank]
    for tag, runs in (("plain", plain), ("inference", inf)):
        for it, a in enumerate(runs):
            assert np.array_equal(torch_to_bits(a[4]), llc), (tag, it, "low-latency combine differs from the oracle")
    names = ("recv_x", "recv_x_scales", "combined", "ll_recv_x", "ll_combined", "fused")
    for it, (a, b) in enumerate(zip(plain, inf)):
        for name, u, v in zip(names, a, b):
            assert u.shape == v.shape and torch.equal(u, v), (it, name, u.shape, v.shape, int((u != v).sum()) if u.shape == v.shape else -1)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: quant_mode="pertoken_fp8_e4m3" through deep_ep.Buffer (normal + low-latency dispatch): returned dtypes / shapes as the
# reference's Ascend950 build returns them (deep_ep.cpp:352-355, buffer.py:652-662), payload bits and scales == oracle
# ----------------------------------------------------------------------------------------------
def gpu_fp8_worker(rank, world, port, cfg):
    run_guarded(_gpu_fp8, rank, world, port, cfg)


def _gpu_fp8(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits
    _set_device(rank)
    W, T, H, K, E, drop = cfg
    group = _init(rank, world, port)
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(512 << 20))
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    xs, idxs, ws = make_inputs(W, T, H, K, E, drop, seed=21)
    x, ti = bits_to_torch(xs[rank]).cuda(), torch.from_numpy(idxs[rank]).cuda()
    tw = torch.from_numpy(np.abs(ws[rank])).cuda()
    for transport in (("push", "pull") if W > 1 else ("pull",)):
        buf.runtime.set_dispatch_transport(transport)
        want = O.normal_dispatch(xs, idxs, E, "fp8")[rank]
        per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
        (rx, rs), _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                      num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                      quant_mode="pertoken_fp8_e4m3")
        n = want.total_recv
        assert rx.dtype == torch.float8_e4m3fn and rs.dtype == torch.float32 and rx.shape[1] == H and lst == want.num_recv_tokens_per_expert_list
        assert np.array_equal(rx.view(torch.uint8).cpu().numpy()[:n], want.recv_x[:n])
        assert np.array_equal(rs.cpu().numpy()[:n].view(np.uint32), want.recv_x_scales[:n].view(np.uint32))
        assert np.array_equal(handle[3].cpu().numpy()[:3 * n], want.recv_src_idx[:3 * n])
        # de-quantise the way a consumer would (torch's own float8 -> float conversion), combine, compare with the oracle's round trip
        y = (rx.float() * rs[:, None]).to(torch.bfloat16)
        assert np.array_equal(torch_to_bits(y)[:n], O.per_token_cast_back(want.recv_x, want.recv_x_scales)[:n])
        out, _, _ = buf.combine(y, handle)
        allw = O.normal_dispatch(xs, idxs, E, "fp8")
        ys = [O.per_token_cast_back(w_.recv_x, w_.recv_x_scales) for w_ in allw]
        comb = O.combine(ys, [w_.recv_src_idx for w_ in allw], [w_.total_recv for w_ in allw], idxs, [np.abs(w_) for w_ in ws], E)[rank]
        assert np.array_equal(torch_to_bits(out), comb)
    MT = T + W
    llw = O.low_latency_dispatch(xs, idxs, MT, E, "fp8")[rank]
    (lx, ls), cnt, h, _, hook = buf.low_latency_dispatch(x, ti, MT, E, quant_mode="pertoken_fp8_e4m3")
    hook()
    nn = llw.total
    assert lx.dtype == torch.float8_e4m3fn and ls.shape == (lx.shape[0],) and np.array_equal(cnt.cpu().numpy(), llw.packed_recv_count)
    assert np.array_equal(lx.view(torch.uint8).cpu().numpy()[:nn], llw.packed_recv_x[:nn])
    assert np.array_equal(ls.cpu().numpy()[:nn].view(np.uint32), llw.packed_recv_x_scales[:nn].view(np.uint32))
    assert np.array_equal(h[0].cpu().numpy()[:3 * nn], llw.src_info)
    # the block-scaled modes are not built: a clear error, as the reference gives off its Ascend950 build (deep_ep.cpp:338-343)
    for qmode in ("mx_fp8_e4m3", "mx_fp4_e2m1"):
        try:
            buf.low_latency_dispatch(x, ti, MT, E, quant_mode=qmode)
            raise AssertionError(f"{qmode} must be rejected")
        except (RuntimeError, ValueError) as e:
            assert "not supported" in str(e)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: a peer that never shows up must surface as a RuntimeError (bounded spins), not a hang
# ----------------------------------------------------------------------------------------------
def gpu_timeout_worker(rank, world, port, cfg):
    run_guarded(_gpu_timeout, rank, world, port, cfg)


def _gpu_timeout(rank, world, port, cfg):
    import time
    import deep_ep
    _set_device(rank)
    os.environ["DEEPEP_TIMEOUT_MS"] = "300"
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(256 << 20))
    group = _init(rank, world, port, "gloo")
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    T, H, K, E = 16, 512, 2, 8
    x = torch.randn((T, H), device="cuda").to(torch.bfloat16)
    ti = torch.randint(0, E, (T, K), device="cuda")
    tw = torch.rand((T, K), device="cuda")
    if rank == 0:
        t0 = time.time()
        try:
            per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
            buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in, num_tokens_per_expert=per_expert, topk_idx=ti,
                         topk_weights=tw)
            torch.cuda.synchronize()
            raised = False
        except RuntimeError as e:
            raised = True
            assert "imeout" in str(e) or "status" in str(e) or "never arrived" in str(e), str(e)
        assert raised, "dispatch without its peer must raise"
        assert time.time() - t0 < 30
    else:
        time.sleep(3.0)          # never joins the exchange
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: long-sequence mode (DEEPEP_NORMAL_LONG_SEQ_*): several slices must give exactly the single-shot result
# ----------------------------------------------------------------------------------------------
def gpu_long_seq_worker(rank, world, port, cfg):
    run_guarded(_gpu_long_seq, rank, world, port, cfg)


def _gpu_long_seq(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits
    _set_device(rank)
    W, T, H, K, E, drop, quant, rounds, per_round = cfg
    os.environ["DEEPEP_NORMAL_LONG_SEQ_ROUND"] = str(rounds)
    os.environ["DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS"] = str(per_round)
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(768 << 20))
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "20000")
    group = _init(rank, world, port, "gloo")
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    xs, idxs, ws = make_inputs(W, T, H, K, E, drop, seed=321)
    x = bits_to_torch(xs[rank]).cuda()
    ti = torch.from_numpy(idxs[rank]).cuda()
    tw = torch.from_numpy(ws[rank]).cuda()
    want = O.normal_dispatch(xs, idxs, E, quant)           # the single-shot oracle is the contract
    per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti, E)
    recv_x, _, _, lst, handle, _ = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                num_tokens_per_expert=per_expert, topk_idx=ti, topk_weights=tw,
                                                quant_mode="int8" if quant else None)
    n = want[rank].total_recv
    assert lst == want[rank].num_recv_tokens_per_expert_list, (lst, want[rank].num_recv_tokens_per_expert_list)
    assert np.array_equal(handle[3].cpu().numpy()[:3 * n], want[rank].recv_src_idx[:3 * n])
    assert np.array_equal(handle[5].cpu().numpy(), want[rank].send_head)
    if quant:
        assert np.array_equal(recv_x[0].cpu().numpy()[:n], want[rank].recv_x[:n])
        assert np.array_equal(recv_x[1].cpu().numpy()[:n], want[rank].recv_x_scales[:n])
        y = bits_to_torch(O.per_token_cast_back(want[rank].recv_x, want[rank].recv_x_scales)).cuda()
    else:
        assert np.array_equal(torch_to_bits(recv_x)[:n], want[rank].recv_x[:n])
        y = recv_x
    ys = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
    comb_want = O.combine(ys, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)
    out, _, _ = buf.combine(y, handle)
    assert np.array_equal(torch_to_bits(out), comb_want[rank]), "long-sequence combine mismatch"
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# CPU / gloo: the comm-shaped CPU baseline (oracle/cpu_alltoall.py) against the oracle
# ----------------------------------------------------------------------------------------------
def cpu_baseline_worker(rank, world, port, cfg):
    run_guarded(_cpu_baseline, rank, world, port, cfg)


def _cpu_baseline(rank, world, port, cfg):
    from oracle import cpu_alltoall as CA
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits, bf16_bits_to_f32
    W, T, H, K, E, drop, quant = cfg
    group = _init(rank, world, port)
    xs, idxs, ws = make_inputs(W, T, H, K, E, drop, seed=77)
    x, ti, tw = bits_to_torch(xs[rank]), torch.from_numpy(idxs[rank]), torch.from_numpy(ws[rank])
    out, recv, scale, lay = CA.one_pass(x, ti, tw, E, group, quant)
    want = O.normal_dispatch(xs, idxs, E, quant)
    n = want[rank].total_recv
    assert recv.shape[0] == n and lay["per_expert"].tolist() == want[rank].num_recv_tokens_per_expert_list
    if quant:      # same receive order (local expert, source rank, row-major (t, k)) and the same INT8 bits as the oracle
        assert np.array_equal(recv.numpy(), want[rank].recv_x[:n])
        assert np.array_equal(scale.numpy().view(np.uint32), want[rank].recv_x_scales[:n].view(np.uint32))
    else:
        assert np.array_equal(torch_to_bits(recv), want[rank].recv_x[:n])
    ys = [O.per_token_cast_back(w.recv_x, w.recv_x_scales) if quant else w.recv_x for w in want]
    comb = O.combine(ys, [w.recv_src_idx for w in want], [w.total_recv for w in want], idxs, ws, E)[rank]
    a, b = bf16_bits_to_f32(torch_to_bits(out)), bf16_bits_to_f32(comb)
    # the baseline accumulates in expert order, the reference kernel in k order: same terms, different fp32 rounding
    assert O.calc_diff(a, b) < 1e-6
    assert np.abs(a - b).max() <= 2.0 ** -6 * max(np.abs(b).max(), 1e-6)
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: low-latency dispatch / combine / fused_deep_moe captured ONCE in a HIP graph and replayed with fresh inputs
# (the call epoch and ping-pong half live on the device: reference cam_moe_dispatch_normal.h:273-286)
# ----------------------------------------------------------------------------------------------
def gpu_graph_worker(rank, world, port, cfg):
    run_guarded(_gpu_graph, rank, world, port, cfg)


def _gpu_graph(rank, world, port, cfg):
    import faulthandler
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits, bf16_bits_to_f32
    faulthandler.dump_traceback_later(200, exit=True)
    _set_device(rank)
    W, T, H, I, K, E, replays = cfg
    L = E // W
    group = _init(rank, world, port)
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(512 << 20))
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "20000")
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    rng = np.random.default_rng(11)
    w13 = [rng.integers(-16, 16, (L, 2 * I, H)).astype(np.int8) for _ in range(W)]
    w2 = [rng.integers(-16, 16, (L, H, I)).astype(np.int8) for _ in range(W)]
    s13 = [(rng.random((L, 2 * I)) * 4e-4 + 1.5e-3).astype(np.float32) for _ in range(W)]
    s2 = [(rng.random((L, H)) * 4e-4 + 1.5e-3).astype(np.float32) for _ in range(W)]
    perm = O.permute_fusion_cols(2 * I)
    w13_d = torch.from_numpy(np.ascontiguousarray(w13[rank][:, perm, :])).cuda()
    s13_d = torch.from_numpy(np.ascontiguousarray(s13[rank][:, perm])).cuda()
    w2_d, s2_d = torch.from_numpy(w2[rank]).cuda(), torch.from_numpy(s2[rank]).cuda()

    def inputs(seed):
        xs, idxs, _ = make_inputs(W, T, H, K, E, 0.1, seed=seed)
        xs, idxs = [x[:T] for x in xs], [i[:T] for i in idxs]
        ws = [np.abs(np.random.default_rng(seed + r).standard_normal((T, K))).astype(np.float32) for r in range(W)]
        return xs, idxs, ws

    x_s = torch.zeros((T, H), dtype=torch.bfloat16, device="cuda")
    ti_s = torch.zeros((T, K), dtype=torch.int64, device="cuda")
    tw_s = torch.zeros((T, K), dtype=torch.float32, device="cuda")

    def load(seed):
        xs, idxs, ws = inputs(seed)
        x_s.copy_(bits_to_torch(xs[rank]).cuda())
        ti_s.copy_(torch.from_numpy(idxs[rank]).cuda())
        tw_s.copy_(torch.from_numpy(ws[rank]).cuda())
        return xs, idxs, ws

    def step():
        (rx, rs), cnt, handle, _, _ = buf.low_latency_dispatch(x_s, ti_s, T, E, use_fp8=True)
        y = (rx.float() * rs[:, None]).to(torch.bfloat16)
        out, _, _ = buf.low_latency_combine(y, ti_s, tw_s, handle)
        fused, rc = buf.fused_deep_moe(x_s, ti_s, tw_s, w13_d, s13_d, w2_d, s2_d, T, E)
        # normal mode in DeepEP's graph-friendly form (num_worst_tokens: worst-case sized outputs, no host sync) + combine
        per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(ti_s, E)
        (nrx, nrs), _, _, lst, nh, _ = buf.dispatch(x_s, num_tokens_per_rank=per_rank, is_token_in_rank=is_in,
                                                    num_tokens_per_expert=per_expert, topk_idx=ti_s, topk_weights=tw_s,
                                                    quant_mode="int8", num_worst_tokens=T * K * W)
        assert lst == []
        ny = (nrx.float() * nrs[:, None]).to(torch.bfloat16)      # rows past the received total are garbage and never pushed
        nout, _, _ = buf.combine(ny, nh)
        return rx, rs, cnt, handle, out, fused, rc, nrx, nrs, nh, nout

    # warm-up on a side stream (allocator pools, one-off function attributes), an ODD number of calls so that the captured
    # graph is first replayed on the other ping-pong half than the one it would have been captured "for" by a host-side counter
    load(1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    dist.barrier()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rx, rs, cnt, handle, out, fused, rc, nrx, nrs, nh, nout = step()
    for rep in range(replays):
        xs, idxs, ws = load(100 + rep)
        torch.cuda.synchronize()
        dist.barrier()
        g.replay()
        torch.cuda.synchronize()
        llw = O.low_latency_dispatch(xs, idxs, T, E, True)
        nn = llw[rank].total
        assert np.array_equal(cnt.cpu().numpy(), llw[rank].packed_recv_count), rep
        assert np.array_equal(handle[1].cpu().numpy(), llw[rank].layout_range), rep
        assert np.array_equal(handle[0].cpu().numpy()[:3 * nn], llw[rank].src_info), rep
        assert np.array_equal(rx.cpu().numpy()[:nn], llw[rank].packed_recv_x[:nn]), rep
        assert np.array_equal(rs.cpu().numpy()[:nn].view(np.uint32), llw[rank].packed_recv_x_scales[:nn].view(np.uint32)), rep
        yls = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) for w in llw]
        want = O.combine(yls, [w.src_info for w in llw], [w.total for w in llw], idxs, ws, E)
        assert np.array_equal(torch_to_bits(out), want[rank]), f"replay {rep}: LL combine mismatch"
        fw = O.fused_deep_moe(xs, idxs, ws, w13, s13, w2, s2, T, E)[rank]
        got, ref = bf16_bits_to_f32(torch_to_bits(fused)), bf16_bits_to_f32(fw)
        assert np.array_equal(rc.cpu().numpy(), llw[rank].layout_range), rep
        assert O.calc_diff(got, ref) < 1e-5, (rep, O.calc_diff(got, ref))
        assert np.mean(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-2)) < 4e-4, rep
        nw = O.normal_dispatch(xs, idxs, E, True)
        n = nw[rank].total_recv
        assert np.array_equal(nrx.cpu().numpy()[:n], nw[rank].recv_x[:n]), rep
        assert np.array_equal(nrs.cpu().numpy()[:n].view(np.uint32), nw[rank].recv_x_scales[:n].view(np.uint32)), rep
        assert np.array_equal(nh[3].cpu().numpy()[:3 * n], nw[rank].recv_src_idx[:3 * n]), rep
        nys = [O.per_token_cast_back(w_.recv_x, w_.recv_x_scales) for w_ in nw]
        ncomb = O.combine(nys, [w_.recv_src_idx for w_ in nw], [w_.total_recv for w_ in nw], idxs, ws, E)
        assert np.array_equal(torch_to_bits(nout), ncomb[rank]), f"replay {rep}: normal combine mismatch"
    faulthandler.cancel_dump_traceback_later()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# tensor-parallel RMSNorm + RoPE: the wrapper's all-reduce between its two launches, two column shards on one GPU (gloo carries the
# device tensor through the host; RCCL would refuse two ranks on one device)
# ----------------------------------------------------------------------------------------------
def gpu_tp_rmsnorm_worker(rank, world, port, cfg):
    run_guarded(_gpu_tp_rmsnorm, rank, world, port, cfg)


def _gpu_tp_rmsnorm(rank, world, port, cfg):
    import torch
    import torch.distributed as dist
    from oracle import kernels as OK
    from sgl_kernel_npu.norm.split_qkv_tp_rmsnorm_rope import split_qkv_tp_rmsnorm_rope
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _set_device(rank)
    B, qh, kvh, hd = cfg                                   # per-rank shard sizes
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(1234)                # the same full tensors on every rank
    full_q, full_k, full_v = (torch.randn(B, n * world, generator=g).to(dt) for n in (qh, kvh, kvh))
    fqw, fkw = torch.randn(qh * world, generator=g).to(dt), torch.randn(kvh * world, generator=g).to(dt)
    cos, sin = torch.rand(B, hd, generator=g).to(dt), torch.rand(B, hd, generator=g).to(dt)
    sl = lambda t, n: t[..., rank * n:(rank + 1) * n]
    shard = torch.cat([sl(full_q, qh), sl(full_k, kvh), sl(full_v, kvh)], dim=-1).contiguous()
    q, k, v = split_qkv_tp_rmsnorm_rope(shard.cuda(), cos.cuda(), sin.cuda(), qh, kvh, hd, 1e-6, sl(fqw, qh).contiguous().cuda(),
                                        sl(fkw, kvh).contiguous().cuda(), hd, world, dist.group.WORLD)
    # what the all-reduce must have supplied: the other shards' local means of squares
    other = torch.zeros(B, 2)
    for r in range(world):
        if r != rank:
            other[:, 0] += full_q[:, r * qh:(r + 1) * qh].float().pow(2).mean(-1)
            other[:, 1] += full_k[:, r * kvh:(r + 1) * kvh].float().pow(2).mean(-1)
    wq, wk, wv = OK.split_qkv_tp_rmsnorm_rope(shard, cos, sin, qh, kvh, hd, 1e-6, sl(fqw, qh), sl(fkw, kvh), hd, tp_world=world, other_var=other)
    assert torch.equal(v.cpu(), wv)
    assert torch.allclose(q.cpu().float(), wq.float(), rtol=2 ** -6, atol=2 ** -6), (q.cpu().float() - wq.float()).abs().max()
    assert torch.allclose(k.cpu().float(), wk.float(), rtol=2 ** -6, atol=2 ** -6)
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# two get_dispatch_layout calls in flight on different streams of ONE Buffer (two-batch overlap): each cooperative launch
# borrows its own pair of sync words from the Buffer's ring, so neither leaves the grid barrier on the other's arrivals
# ----------------------------------------------------------------------------------------------
def gpu_layout_two_streams_worker(rank, world, port, cfg):
    run_guarded(_gpu_layout_two_streams, rank, world, port, cfg)


def _gpu_layout_two_streams(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    _set_device(rank)
    group = _init(rank, world, port)
    T, K, E, rounds = cfg
    buf = deep_ep.Buffer(group, low_latency_mode=False)
    rng = np.random.default_rng(11)
    idx = [make_topk(rng, T + 256 * i, K, E, 0.1) for i in range(2)]
    want = [O.dispatch_layout(i, E, world) for i in idx]
    dev = [torch.from_numpy(i).cuda() for i in idx]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rnd in range(rounds):
        got = [None, None]
        for s in (0, 1):
            with torch.cuda.stream(streams[s]):
                for _ in range(4):                      # several launches queued per stream: the two streams' kernels interleave
                    got[s] = buf.get_dispatch_layout(dev[s], E)
        torch.cuda.synchronize()
        for s in (0, 1):
            per_rank, _, per_expert, is_in, _ = got[s]
            assert np.array_equal(per_expert.cpu().numpy(), want[s]["num_tokens_per_expert"]), (rnd, s)
            assert np.array_equal(per_rank.cpu().numpy(), want[s]["num_tokens_per_rank"]), (rnd, s)
            assert np.array_equal(is_in.cpu().numpy().astype(bool), want[s]["is_token_in_rank"].astype(bool)), (rnd, s)
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# GPU: shared-expert ranks (MOE_SHARED_EXPERT_RANK_NUM = S): low-latency dispatch / combine bit for bit against the kernel restatement,
# fused_deep_moe at the reference bar
# ----------------------------------------------------------------------------------------------
def gpu_shared_expert_worker(rank, world, port, cfg):
    run_guarded(_gpu_shared_expert, rank, world, port, cfg)


def _gpu_shared_expert(rank, world, port, cfg):
    import deep_ep
    from oracle import ep as O
    from oracle.bf16 import bits_to_torch, torch_to_bits, bf16_bits_to_f32
    _set_device(rank)
    W, S, T, H, K, E, drop, quant, I = cfg
    os.environ["MOE_SHARED_EXPERT_RANK_NUM"] = str(S)
    group = _init(rank, world, port)
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(768 << 20))
    os.environ.setdefault("DEEPEP_TIMEOUT_MS", "20000")
    buf = deep_ep.Buffer(group, low_latency_mode=True)
    L = E // (W - S)
    nl = 1 if rank < S else L
    for it in range(2):
        xs, idxs, ws = make_inputs(W, T, H, K, E, drop, seed=300 + it)
        if drop > 0:
            idxs[W - 1][0, :] = -1            # a token without any active selection: it does not go to the shared expert either
        x = bits_to_torch(xs[rank]).cuda()
        ti = torch.from_numpy(idxs[rank]).cuda()
        if it == 1:
            ti = ti.int()
        MT = T + W
        llw = O.low_latency_dispatch_shared(xs, idxs, MT, E, quant, S)
        rx, cnt, h, _, hook = buf.low_latency_dispatch(x, ti, MT, E, use_fp8=quant)
        hook()
        me = llw[rank]
        # shapes of the reference host (deep_ep.cpp:866-874,914-917): one local expert and global_bs / S rows on a shared rank
        assert cnt.shape == (nl,) and h[1].shape == (nl * W,)
        assert (rx[0] if quant else rx).shape[0] == (MT * W // S if rank < S else MT * W * min(K, L))
        assert np.array_equal(cnt.cpu().numpy(), me.packed_recv_count), (cnt.cpu().numpy(), me.packed_recv_count)
        assert np.array_equal(h[1].cpu().numpy(), me.layout_range)
        nn = me.total
        assert np.array_equal(h[0].cpu().numpy()[:3 * nn], me.src_info)
        if quant:
            assert np.array_equal(rx[0].cpu().numpy()[:nn], me.packed_recv_x[:nn])
            assert np.array_equal(rx[1].cpu().numpy()[:nn], me.packed_recv_x_scales[:nn])
            yl = bits_to_torch(O.per_token_cast_back(me.packed_recv_x, me.packed_recv_x_scales)).cuda()
        else:
            assert np.array_equal(torch_to_bits(rx)[:nn], me.packed_recv_x[:nn])
            yl = rx
        yls = [O.per_token_cast_back(w.packed_recv_x, w.packed_recv_x_scales) if quant else w.packed_recv_x for w in llw]
        wabs = [np.abs(w_) for w_ in ws]
        want = O.low_latency_combine_shared(yls, [w.src_info for w in llw], [w.total for w in llw], idxs, wabs, E)
        outl, _, hook = buf.low_latency_combine(yl, ti, torch.from_numpy(wabs[rank]).cuda(), h)
        hook()
        assert np.array_equal(torch_to_bits(outl), want[rank]), "LL combine with a shared expert mismatch"
    if I:
        rng = np.random.default_rng(11)
        xs, idxs, _ = make_inputs(W, T, H, K, E, drop, seed=9)
        xs = [x[:T] for x in xs]
        idxs = [i[:T] for i in idxs]
        ws = [np.abs(rng.standard_normal((T, K))).astype(np.float32) for _ in range(W)]
        nls = [1 if r < S else L for r in range(W)]
        w13 = [rng.integers(-16, 16, (n, 2 * I, H)).astype(np.int8) for n in nls]
        w2 = [rng.integers(-16, 16, (n, H, I)).astype(np.int8) for n in nls]
        s13 = [(rng.random((n, 2 * I)) * 4e-4 + 1.5e-3).astype(np.float32) for n in nls]
        s2 = [(rng.random((n, H)) * 4e-4 + 1.5e-3).astype(np.float32) for n in nls]
        want = O.fused_deep_moe(xs, idxs, ws, w13, s13, w2, s2, T, E, shared_expert_rank_num=S)[rank]
        perm = O.permute_fusion_cols(2 * I)
        w13_p = torch.from_numpy(np.ascontiguousarray(w13[rank][:, perm, :])).cuda()
        s13_p = torch.from_numpy(np.ascontiguousarray(s13[rank][:, perm])).cuda()
        x = bits_to_torch(xs[rank]).cuda()
        ti = torch.from_numpy(idxs[rank]).cuda()
        tw = torch.from_numpy(ws[rank]).cuda()
        ll = O.low_latency_dispatch_shared(xs, idxs, T, E, True, S)[rank]
        for _ in range(2):
            out, ep_recv_count = buf.fused_deep_moe(x, ti, tw, w13_p, s13_p, torch.from_numpy(w2[rank]).cuda(), torch.from_numpy(s2[rank]).cuda(), T, E)
        assert np.array_equal(ep_recv_count.cpu().numpy(), ll.layout_range)
        got = bf16_bits_to_f32(torch_to_bits(out))
        ref = bf16_bits_to_f32(want)
        assert O.calc_diff(got, ref) < 1e-5, O.calc_diff(got, ref)
        denom = np.maximum(np.abs(ref), 1e-2)
        assert np.mean(np.abs(got - ref) / denom) < 4e-4, np.mean(np.abs(got - ref) / denom)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------
# CPU / gloo: the host side of the start-up check of the two-launch low-latency forms (deep_ep/buffer.py::_check_in_launch_handoff)
# ----------------------------------------------------------------------------------------------
def cpu_handoff_check_worker(rank, world, port, cfg):
    run_guarded(_cpu_handoff_check, rank, world, port, cfg)


def _cpu_handoff_check(rank, world, port, cfg):
    import warnings
    import deep_ep
    failing_rank, raises = cfg
    group = _init(rank, world, port)

    class Stub:                      # what the check needs of deep_ep_cpp.Buffer
        def __init__(self):
            self.forms = None
            self.args = None

        def self_test_in_launch(self, timeout_ms, skip_from):
            self.args = (timeout_ms, skip_from)
            if raises and rank == failing_rank:
                raise RuntimeError("device lost")
            return rank != failing_rank

        def set_two_launch_forms(self, ok):
            self.forms = ok

    buf = object.__new__(deep_ep.Buffer)
    buf.runtime, buf.rank, buf.group, buf.group_size = Stub(), rank, group, world

    def agree(ok):
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok), group=group)
        return all(flags)

    os.environ.pop("DEEPEP_SELF_TEST_STALE_RANK", None)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        buf._check_in_launch_handoff(agree)
    # one rank failing (or raising) switches EVERY rank to the three-launch forms, and every rank says so
    assert buf.runtime.forms is (failing_rank is None)
    assert buf.runtime.args[1] == -1
    said = [str(c.message) for c in caught]
    assert (failing_rank is None) == (not any("three-launch forms" in m for m in said)), said
    if failing_rank is not None:
        assert any(("on this rank" in m) == (rank == failing_rank) for m in said if "three-launch forms" in m), said
    # the test hook reaches the named rank only
    os.environ["DEEPEP_SELF_TEST_STALE_RANK"] = "1"
    buf.runtime = Stub()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        buf._check_in_launch_handoff(agree)
    assert buf.runtime.args[1] == (1 if rank == 1 else -1)
    dist.barrier()
    dist.destroy_process_group()
