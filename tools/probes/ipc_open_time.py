"""Probe: how long does mapping the peers' windows (hipIpcOpenMemHandle inside deep_ep.Buffer.__init__) take as a function of
world size and window size, with all ranks on ONE GPU?  usage: ipc_open_time.py W window_MiB"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, W, mib, port):
    import faulthandler
    faulthandler.dump_traceback_later(int(os.getenv("PROBE_TIMEOUT", "25")), exit=True)
    sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["DEEPEP_WINDOW_BYTES"] = str(mib << 20)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    import deep_ep
    t0 = time.time()
    buf = deep_ep.Buffer(dist.group.WORLD)
    dt = time.time() - t0
    print(f"W={W} window={mib}MiB rank {rank}: Buffer() {dt:.2f}s p2p={buf.p2p_available}", flush=True)
    dist.barrier()
    del buf
    dist.destroy_process_group()


if __name__ == "__main__":
    W, mib = int(sys.argv[1]), int(sys.argv[2])
    mp.spawn(worker, args=(W, mib, 29400 + W + mib % 97), nprocs=W, join=True)
