"""Scratch: the two-launch low-latency forms at larger sizes than the test matrix (rank r brings T + r tokens), bit-exact against the oracle."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import mp_workers
from test_deep_ep_gpu import _spawn
if __name__ == "__main__":
    os.environ.setdefault("DEEPEP_WINDOW_BYTES", str(3 << 30))
    # (several ranks on ONE GPU: the Buffer caps the waiting launches at 64 workgroups when the bootstrap finds ranks sharing a device --
    #  with 512, two ranks at a few hundred tokens timed out in half of the runs, dispatch.hip: mi_ep_ll_wait_pack)
    for cfg in [(1, 1000, 7168, 8, 32, True, ("2", "2")), (2, 500, 7168, 8, 32, True, ("2", "2")), (2, 300, 2048, 8, 64, False, ("2", "2")),
                (1, 700, 2048, 8, 64, False, ("2", "2")), (2, 1020, 1024, 6, 32, True, ("2", "2")), (4, 60, 2048, 8, 64, False, ("2", "2"))]:
        _spawn(mp_workers.gpu_ll_empty_rank_worker, cfg[0], cfg)
        print("ok", cfg, flush=True)
