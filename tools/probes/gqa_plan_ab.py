"""Scratch: decode_gqa, planned (device-built work list) against uniform splits, Llama-70B-shaped decode (64 / 8 heads, d = 128), full and ragged."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
page, D = 64, 128
for B, Hq, Hkv, S in ((4, 64, 8, 32768), (16, 64, 8, 8192), (64, 64, 8, 4096), (256, 64, 8, 4096)):
    g = torch.Generator(device="cuda").manual_seed(1)
    maxp = S // page
    nb = B * maxp
    q = torch.randn((B, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
    k = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
    v = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
    bt = torch.randperm(nb, generator=g, device="cuda").to(torch.int32).reshape(B, maxp)
    out = torch.empty((B, Hq, D), dtype=torch.bfloat16, device="cuda")
    full = torch.full((B,), S, dtype=torch.int32, device="cuda")
    rag = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32)

    def t(ls, n, reps=30):
        call = lambda: torch.ops.npu.decode_gqa(q, k, v, out, ls, D ** -0.5, page, bt, n)
        for _ in range(10):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            call()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    for name, ls in (("full", full), ("ragged", rag)):
        byts = float(ls.sum().item()) * Hkv * D * 2 * 2
        row = {n: t(ls, n) for n in (-1, 0, 1, 2, 4, 8)}
        print(f"B={B} {name}:", " ".join(f"{'planned' if n < 0 else ('library' if n == 0 else f'{n} splits')} {us:.1f} us ({byts / us / 1e6:.2f} TB/s)" for n, us in row.items()), flush=True)
    del q, k, v, bt, out
