"""Scratch: planned (device-built work list) against uniform splits for the C4 batch (AB_HQ heads per rank: a TP shard), full-length and ragged, one process, alternating."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.bench_hooks import _mla_inputs

B, Hq, S, page = int(os.environ.get("AB_B", 128)), int(os.environ.get("AB_HQ", 16)), int(os.environ.get("AB_S", 4096)), 64
q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page)
_, _, _, _, rlens = _mla_inputs(B, Hq, S, page, ragged=True)
g = torch.Generator(device="cuda").manual_seed(3)
skew = torch.randint(1, 400, (B,), generator=g, device="cuda").to(torch.int32)
skew[::16] = S                                     # a few long sequences among short ones
out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
call = lambda ls, n: torch.ops.npu.decode_mla(q, kn, kr, out, ls, 576 ** -0.5, page, bt, n)
for _ in range(300):
    call(lens, 0)
torch.cuda.synchronize()


def t(ls, n, reps=50):
    for _ in range(10):
        call(ls, n)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call(ls, n)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for name, ls in (("full", lens), ("ragged", rlens), ("skewed", skew)):
    byts = float(ls.sum().item()) * 1152 + B * Hq * 2176
    for rnd in range(2):
        row = {n: t(ls, n) for n in (0, 1, 2, 4, 16, 64)}
        print(name, "round", rnd, " ".join(f"{'planned' if n == 0 else f'{n} splits'}: {us:.1f} us ({byts / us / 8e6:.3f})" for n, us in row.items()), flush=True)
