"""Scratch: host-side time of the deep_ep calls of one normal-mode step (GPU idle at every call: pure launch-path cost)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import deep_ep
import bench
x, idx, w = bench.make_inputs(0, 4096)
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
out, n, y, recv, handle = bench.one_step(buf, x, idx, w, None)
for _ in range(5): bench.one_step(buf, x, idx, w, y)
torch.cuda.synchronize()
acc = {"layout": 0.0, "dispatch": 0.0, "combine": 0.0}
N = 30
for _ in range(N):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    per_rank, _, per_expert, is_in, _ = buf.get_dispatch_layout(idx, bench.EXPERTS)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t1b = time.perf_counter()
    r = buf.dispatch(x, num_tokens_per_rank=per_rank, is_token_in_rank=is_in, num_tokens_per_expert=per_expert, topk_idx=idx,
                     topk_weights=w, quant_mode="int8")
    t2 = time.perf_counter(); torch.cuda.synchronize(); t2b = time.perf_counter()
    o = buf.combine(y, r[4])
    t3 = time.perf_counter()
    acc["layout"] += t1 - t0; acc["dispatch"] += t2 - t1b; acc["combine"] += t3 - t2b
print({k: round(v / N * 1e6, 1) for k, v in acc.items()}, "us per call on the host (dispatch includes its own wait for the count exchange: stage + notify ~45 us of GPU time)")
dist.destroy_process_group()
