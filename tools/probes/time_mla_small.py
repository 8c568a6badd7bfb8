"""Scratch: MLA paged decode with kv groups of 16 / 64 heads (the TP-sharded shapes; mla_decode_kernel, not the wide one)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs
from sgl_kernel_npu.attention.decode_attention import decode_mla

for B, Hq in ((128, 16), (128, 64), (32, 16)):
    S, page = 4096, 64
    q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    sm = 576 ** -0.5
    for _ in range(50):
        decode_mla(q, kn, kr, out, lens, sm, page, bt)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        decode_mla(q, kn, kr, out, lens, sm, page, bt)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    byts = float(lens.sum().item()) * 576 * 2 + B * Hq * (576 + 512) * 2
    print(f"B={B} Hq={Hq}: {us:.1f} us  {byts / us / 1e3:.0f} GB/s", flush=True)
