"""Scratch: dispatch layout at C2 size (4096 tokens, top-8 of 256, 8 ranks): three launches vs the cooperative single launch."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import ep_harness as Hh
from capi import ptr, stream_ptr
for T in (1024, 1025, 4096, 16384):
    E, W, K = 256, 8, 8
    idx = torch.topk(torch.rand((T, E), device="cuda"), K, dim=-1)[1]
    i32 = dict(dtype=torch.int32, device="cuda")
    o = [torch.empty(W, **i32), torch.empty(E, **i32), torch.empty((T, W), **i32), torch.empty((T, K), **i32), torch.empty(E, **i32)]
    wsb = Hh.lib().mi_ep_dispatch_layout_workspace(T, K, E)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    words = torch.zeros(2, **i32)
    args = [ptr(idx), 0, T, K, E, W] + [ptr(t) for t in o] + [ptr(ws), wsb]
    for coop in (False, True):
        call = lambda: Hh.lib().mi_ep_dispatch_layout(*args, ptr(words) if coop else None, stream_ptr())
        for _ in range(20): call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): call()
        b.record(); torch.cuda.synchronize()
        print(f"T={T} {'one launch  ' if coop else 'three launches'}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (back to back, outputs preallocated)", flush=True)
