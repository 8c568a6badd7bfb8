"""GPU tests of the full boundary: deep_ep.Buffer -> deep_ep_cpp (C++ runtime) -> HIP kernels, with W ranks as W
processes sharing ONE GPU and their windows mapped through hipIpc (the same code path 8 GPUs take over xGMI)."""
import pytest

import mp_workers
from test_deep_ep_plumbing_cpu import _spawn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    # W, T, H, K, E, drop, quant, strategy, iterations
    (1, 256, 1024, 2, 8, 0.0, False, "default", 2),      # BASELINE C1 on a GPU
    (1, 64, 7168, 8, 256, 0.1, True, "default", 2),
    (2, 48, 512, 4, 16, 0.2, True, "default", 3),
    (4, 33, 7168, 8, 256, 0.0, True, "default", 3),
    (4, 40, 1024, 8, 64, 0.3, False, "default", 2),
    (8, 16, 2048, 8, 256, 0.1, True, "default", 3),
    (2, 1200, 512, 4, 16, 0.0, True, "default", 3),       # 300 tokens, then 1200 twice: speculative receive miss, then hit
    (1, 64, 1024, 4, 16, 0.1, True, "alltoall", 1),
])
def test_buffer_multi_process_one_gpu(cfg):
    _spawn(mp_workers.gpu_buffer_worker, cfg[0], cfg)


def test_c2_size_eight_processes():
    """BASELINE C2 (EP = 8, 4096 tok/rank, H = 7168, top-8 of 256, INT8 dispatch / BF16 combine) through deep_ep.Buffer, eight
    processes with hipIpc-mapped windows, both dispatch transports: counts, receive order, sampled payload bits, round trip."""
    _spawn(mp_workers.gpu_c2_size_worker, 8, (8, 4096, 7168, 8, 256))


@pytest.mark.parametrize("cfg", [
    # W, T, H, I, K, E, weight layout
    (1, 24, 512, 128, 4, 8, "native"),
    (2, 40, 512, 128, 4, 8, "reference"),     # (square H == I would be ambiguous: native layout is assumed)
    (4, 16, 1024, 128, 8, 32, "native"),
    (1, 64, 7168, 2048, 8, 8, "native"),          # DeepSeek-V3 hidden / intermediate, 8 local experts
    (2, 24, 512, 256, 4, 8, "ffn"),               # FuseMode.DISPATCH_FFN_COMBINE: plain weights, int64 scale bits
    (1, 32, 1024, 384, 8, 16, "ffn"),
    # BASELINE C5 shapes (DeepSeek-V3 hidden 7168 / 2I = 4096) with 1024 rows per local expert: rows_hint > 96, i.e. the
    # 256 x 256 x 64 workgroup tile, the LDS-transposed epilogue and the multi-tile lookup run end to end behind deep_ep.Buffer
    (1, 1024, 7168, 2048, 8, 8, "native"),
    (2, 512, 7168, 2048, 8, 8, "native"),         # same through two processes (windows mapped with hipIpc)
    # more than 512 tokens per rank: the prefill-size dispatch leg (normal-mode exchange without host sync, worst-case sized
    # buffers, pooled GEMM tile workers) WITH peers -- the branch BASELINE C5 (4096 tok/rank) takes
    (2, 640, 512, 128, 4, 8, "native"),
    (4, 768, 512, 256, 8, 32, "native"),
    (2, 1024, 1024, 256, 8, 16, "reference"),
    (4, 600, 512, 128, 4, 16, "ffn"),
])
def test_fused_deep_moe(cfg):
    _spawn(mp_workers.gpu_fused_moe_worker, cfg[0], cfg)


@pytest.mark.parametrize("cfg", [(1, 1024, 7168, 2048, 8, 8, "native"), (2, 640, 512, 128, 4, 8, "native"), (2, 24, 512, 256, 4, 8, "ffn"),
                                 (4, 16, 1024, 128, 8, 32, "native")])
def test_fused_deep_moe_with_the_requantisation_in_gemm1(cfg):
    """MI_EP_FUSED_REQUANT=1: GEMM1 requantises its rows in its epilogue (no fp32 intermediate, no rowquant launch) -- the reference's
    structure, opt-in here; same checks as test_fused_deep_moe, prefill- and decode-size legs, one to four processes on one GPU."""
    import os
    keep = os.environ.get("MI_EP_FUSED_REQUANT")
    os.environ["MI_EP_FUSED_REQUANT"] = "1"
    try:
        _spawn(mp_workers.gpu_fused_moe_worker, cfg[0], cfg)
    finally:
        if keep is None:
            os.environ.pop("MI_EP_FUSED_REQUANT", None)
        else:
            os.environ["MI_EP_FUSED_REQUANT"] = keep


def test_fused_deep_moe_c5_size_eight_processes():
    """BASELINE C5 through deep_ep.Buffer: eight processes with hipIpc-mapped windows, 4096 tokens per rank, DeepSeek-V3 expert
    shapes, 32 local experts per rank; 256 sampled tokens per rank against the per-token float64 evaluation."""
    _spawn(mp_workers.gpu_fused_c5_worker, 8, (8, 4096, 7168, 2048, 8, 32, 256))


@pytest.mark.parametrize("cfg", [
    # W, T, H, I, K, E, replays
    (2, 24, 512, 128, 4, 8, 3),
    (1, 16, 1024, 128, 8, 16, 3),
])
def test_low_latency_calls_replay_in_a_captured_graph(cfg):
    """low_latency_dispatch + low_latency_combine + fused_deep_moe captured once in torch.cuda.graph, replayed with fresh inputs,
    bit-exact (fused: reference tolerance) against the oracle every time: the call epoch / ping-pong half are device-resident."""
    _spawn(mp_workers.gpu_graph_worker, cfg[0], cfg)


@pytest.mark.parametrize("forms", [("1", "1"), ("0", "0"), ("1", "0"), ("0", "2"), ("2", "2"), ("2", "0")],
                         ids=["tails", "launches", "dispatch_tail_only", "combine_row_flags", "tagged_rows_and_row_flags", "dispatch_tagged_rows_only"])
@pytest.mark.parametrize("cfg", [(2, 24, 512, 128, 4, 8, 2), (4, 17, 1024, 128, 8, 32, 2)])
def test_low_latency_launch_forms(cfg, forms):
    """The count exchange of a low-latency dispatch and the "rows pushed" signal + wait of a combine, each either as the TAIL of the
    launch in front of it (the last workgroup to arrive does it: MI_EP_LL_FUSED_COUNTS / MI_EP_COMBINE_FUSED = 1) or as a launch of its
    own (= 0): same rows, tables and sums, bit-exact against the oracle, eagerly and replayed from a captured graph.  MI_EP_COMBINE_FUSED = 2:
    the two-launch combine whose pushed rows raise their own flag words and whose reduce waits per selection (no exchange in between);
    MI_EP_LL_FUSED_COUNTS = 2: the two-launch dispatch whose rows carry the call's tag and whose packing launch collects counts and rows itself."""
    import os
    keep = {k: os.environ.get(k) for k in ("MI_EP_LL_FUSED_COUNTS", "MI_EP_COMBINE_FUSED")}
    os.environ["MI_EP_LL_FUSED_COUNTS"], os.environ["MI_EP_COMBINE_FUSED"] = forms
    try:
        _spawn(mp_workers.gpu_graph_worker, cfg[0], cfg)
        _spawn(mp_workers.gpu_buffer_worker, cfg[0], (cfg[0], 40, 1024, 8, 64, 0.3, True, "default", 2))
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("forms", [("0", "0"), ("2", "2"), ("1", "1"), ("2", "0"), ("0", "2")],
                         ids=["three_launches", "two_launches", "tails", "tagged_dispatch_three_launch_combine", "three_launch_dispatch_flagged_combine"])
@pytest.mark.parametrize("cfg", [(1, 0, 512, 4, 8, True), (2, 0, 1024, 4, 8, True), (4, 0, 512, 8, 32, False)])      # W, T of rank 0, H, K, E, quant
def test_low_latency_pair_with_a_rank_without_tokens(cfg, forms):
    """Rank r brings T + r tokens, rank 0 none at all: its send launch still posts its (zero) counts and leaves the call's epoch, its combine
    launches still complete the call counter -- in every launch form and both mixed pairs, three calls in a row (both ping-pong halves),
    bit-exact."""
    _spawn(mp_workers.gpu_ll_empty_rank_worker, cfg[0], cfg + (forms,))


@pytest.mark.parametrize("cfg", [(2, 24, 1024, 4, 8, True), (4, 9, 512, 8, 32, False)])
def test_two_launch_forms_uncapped_as_on_a_node(cfg):
    """The forms an 8-GPU node takes by default, taken here the same way: no env, the runtime told that every rank owns its GPU (default forms
    2 / 2 selected by the start-up self-test's second leg, waiting launches not capped at 64 workgroups); the batches are small enough for both
    processes' grids to be resident together.  Plus two extra combines per call on the same handle (they fall back to three launches) and a lone
    combine captured in a graph and replayed three times (captured without its dispatch: three launches as well)."""
    _spawn(mp_workers.gpu_ll_empty_rank_worker, cfg[0], cfg + (None, {"own_gpu": True, "repeat_combine": True, "capture_lone_combine": True}))


@pytest.mark.parametrize("forms", [("2", "2"), None], ids=["forms_forced_by_env", "default_forms"])
def test_failed_in_launch_self_test_falls_back_to_three_launches(forms):
    """One rank fails the second self-test leg (its tags / flags arrive in front of a payload that was not rewritten: codes 7000 + s /
    7500 + s at its consumers): EVERY rank reports three-launch forms -- even against MI_EP_LL_FUSED_COUNTS / MI_EP_COMBINE_FUSED = 2 --,
    the window strategies stay, and the dispatch + combine pair is bit-exact."""
    _spawn(mp_workers.gpu_ll_empty_rank_worker, 2, (2, 5, 512, 4, 8, True, forms, {"stale_rank": 1, "own_gpu": forms is None}))


def test_tagged_wire_form_is_chosen_from_shared_values_only():
    """num_max_dispatch_tokens_per_rank = 1200 with 1100 tokens on rank 0 and 20 on rank 1, two-launch forms asked for: the TAGGED dispatch rows
    need the one-launch layout + send (<= 1024 tokens), which rank 0 cannot take -- so NO rank may wait for tags.  The form follows the shared
    bound (1200 > 1024: plain rows + count exchange everywhere), not each rank's own T; the pair is bit-exact instead of timing out."""
    _spawn(mp_workers.gpu_ll_empty_rank_worker, 2,
           (2, 0, 512, 4, 8, True, ("2", "2"), {"tokens": [1100, 20], "max_tokens": 1200}))


@pytest.mark.parametrize("cfg", [(1, 40, 512, 128, 4, 8), (2, 33, 512, 128, 4, 8)])      # W, T, H, I, K, E
def test_every_call_works_under_inference_mode(cfg):
    _spawn(mp_workers.gpu_inference_mode_worker, cfg[0], cfg)


@pytest.mark.parametrize("cfg", [(1, 64, 7168, 8, 64, 0.1), (2, 40, 1024, 4, 16, 0.2), (4, 33, 2048, 8, 64, 0.0)])   # W, T, H, K, E, drop
def test_pertoken_fp8_e4m3_dispatch_through_buffer(cfg):
    _spawn(mp_workers.gpu_fp8_worker, cfg[0], cfg)


def test_missing_peer_raises_instead_of_hanging():
    _spawn(mp_workers.gpu_timeout_worker, 2, None)


@pytest.mark.parametrize("cfg", [
    # W, T, H, K, E, drop, quant, rounds, tokens per round
    (2, 80, 512, 4, 16, 0.1, True, 3, 32),        # 32 + 32 + 16 tokens
    (4, 70, 1024, 8, 32, 0.0, False, 4, 32),      # last slice empty on every rank
    (1, 200, 512, 2, 8, 0.2, True, 2, 128),
])
def test_long_sequence_rounds_match_single_shot(cfg):
    _spawn(mp_workers.gpu_long_seq_worker, cfg[0], cfg)


@pytest.mark.parametrize("world", [2, 4])
def test_split_qkv_tp_rmsnorm_rope_all_reduces_the_variance_across_ranks(world):
    """norm/split_qkv_tp_rmsnorm_rope.py with tp_world > 1: `world` processes hold the column shards of one row batch; the wrapper's
    dist.all_reduce between its two launches must make every shard normalise with the GLOBAL mean of squares."""
    _spawn(mp_workers.gpu_tp_rmsnorm_worker, world, (9, 512, 128, 64))


def test_layout_calls_on_two_streams_do_not_share_barrier_words():
    """get_dispatch_layout issued concurrently on two streams of one Buffer (4096 + 4352 tokens: both take the cooperative
    one-launch form): every launch borrows its own sync-word pair from the Buffer's ring; all tables match the oracle."""
    _spawn(mp_workers.gpu_layout_two_streams_worker, 1, (4096, 8, 256, 10))


@pytest.mark.parametrize("cfg", [
    # W, S (shared-expert ranks), T, H, K, E (routed experts), drop, quant, I (0: no fused_deep_moe leg)
    (2, 1, 24, 512, 2, 4, 0.0, True, 128),
    (4, 1, 33, 1024, 4, 12, 0.2, True, 0),
    (4, 2, 40, 512, 4, 8, 0.1, False, 128),
    (8, 2, 16, 2048, 8, 48, 0.1, True, 0),
    (8, 4, 12, 512, 7, 16, 0.3, True, 256),
    (2, 1, 600, 512, 2, 4, 0.0, True, 128),       # more than 512 tokens: fused_deep_moe takes the prefill-size exchange branch
])
def test_shared_expert_ranks(cfg):
    """MOE_SHARED_EXPERT_RANK_NUM = S (reference deep_ep.cpp:62,866-874,1219-1220): the first S ranks hold the shared expert and receive
    every active token of the sources with their residue; the others hold E / (W - S) routed experts.  Low-latency dispatch (rows, scales,
    triples with k = K, counts, output shapes) and combine (K weighted rows, then the shared row unweighted) bit for bit against the
    restatement of the kernels; fused_deep_moe with per-rank expert counts at the reference's bar."""
    _spawn(mp_workers.gpu_shared_expert_worker, cfg[0], cfg)
