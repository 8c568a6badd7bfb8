"""CPU / gloo plumbing tests (no GPU): deep_ep.Buffer + the alltoall strategies over torch.distributed with the
oracle-backed test double standing in for the HIP kernels.  Covers BASELINE config C1 (world_size=1, 256 tokens,
hidden=1024, top-2, BF16) and the N>1 path at world_size 2."""
import os
import socket

import pytest
import torch.multiprocessing as mp

import mp_workers


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(fn, world, cfg):
    os.environ["PYTHONPATH"] = os.pathsep.join([mp_workers.ROOT, os.path.join(mp_workers.ROOT, "tests"),
                                                os.path.join(mp_workers.ROOT, "sgl-kernel-npu_amd", "python"),
                                                os.environ.get("PYTHONPATH", "")])
    mp.spawn(fn, args=(world, free_port(), cfg), nprocs=world, join=True)


@pytest.mark.parametrize("cfg", [
    (1, 256, 1024, 2, 8, 0.0, False),     # BASELINE C1
    (2, 24, 128, 2, 8, 0.2, True),
    (2, 16, 64, 4, 4, 0.0, False),
])
def test_alltoall_strategy_over_gloo(cfg):
    _spawn(mp_workers.cpu_alltoall_worker, cfg[0], cfg)


@pytest.mark.parametrize("cfg", [(None, False), (1, False), (0, True)], ids=["passes", "rank1_fails", "rank0_raises"])
def test_in_launch_handoff_check_switches_every_rank(cfg):
    """deep_ep/buffer.py::_check_in_launch_handoff over gloo, world_size 2, with a stub runtime: a pass keeps the two-launch forms, one rank
    failing or raising makes EVERY rank call set_two_launch_forms(False) and warn; DEEPEP_SELF_TEST_STALE_RANK reaches the named rank only."""
    _spawn(mp_workers.cpu_handoff_check_worker, 2, cfg)


def test_strategy_registry_and_errors():
    import deep_ep
    from deep_ep.ep_strategy import StrategyMap, get_normal_strategy, get_low_latency_strategy
    from deep_ep.strategies.normal_strategy import resolve_quant
    import torch
    assert StrategyMap.get_strategy("ALLTOALL") == ("alltoall", "alltoall")
    assert StrategyMap.get_strategy("ops") == ("default", "ops")
    with pytest.raises(ValueError):
        StrategyMap.get_strategy("bogus")
    with pytest.raises(ValueError):
        get_normal_strategy("nope")
    with pytest.raises(ValueError):
        get_low_latency_strategy("nope")
    x = torch.zeros((2, 16), dtype=torch.bfloat16)
    assert resolve_quant(x, None)[1:] == ("bf16", False)
    assert resolve_quant(x, "int8")[1:] == ("int8", True)
    with pytest.raises(ValueError):
        resolve_quant(x, "fp3")
    with pytest.raises(TypeError):
        resolve_quant([x], None)
    assert deep_ep.Buffer.get_dispatch_config(1).num_sms % 2 == 0       # the W=1 entry the reference lacks
    with pytest.raises(AssertionError):
        deep_ep.Buffer.get_dispatch_config(3)
    with pytest.raises(AssertionError):
        deep_ep.Buffer.set_num_sms(3)
    assert deep_ep.Buffer.get_low_latency_rdma_size_hint(128, 7168, 8, 256) == 128
    # long-sequence env knobs: same ranges as the reference (csrc/deepep/deep_ep.cpp:63-90)
    from deep_ep.strategies.normal_strategy import long_seq_rounds
    for k in ("DEEPEP_NORMAL_LONG_SEQ_ROUND", "DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS"):
        os.environ.pop(k, None)
    assert long_seq_rounds() == (1, 8192)
    os.environ["DEEPEP_NORMAL_LONG_SEQ_ROUND"], os.environ["DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS"] = "16", "8192"
    assert long_seq_rounds() == (16, 8192)
    for r, t in (("0", "8192"), ("257", "32"), ("2", "16"), ("2", "9000"), ("32", "8192")):
        os.environ["DEEPEP_NORMAL_LONG_SEQ_ROUND"], os.environ["DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS"] = r, t
        with pytest.raises(ValueError):
            long_seq_rounds()
    for k in ("DEEPEP_NORMAL_LONG_SEQ_ROUND", "DEEPEP_NORMAL_LONG_SEQ_PER_ROUND_TOKENS"):
        os.environ.pop(k, None)


@pytest.mark.parametrize("cfg", [
    (2, 40, 128, 4, 8, 0.2, True),
    (4, 24, 64, 2, 8, 0.0, False),
    (1, 16, 64, 2, 4, 0.1, True),
])
def test_cpu_alltoall_baseline_matches_oracle(cfg):
    """bench.py's comm-shaped CPU baseline (W gloo ranks, reference alltoall strategy restated with torch ops)."""
    _spawn(mp_workers.cpu_baseline_worker, cfg[0], cfg)
