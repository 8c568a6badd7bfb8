"""Scratch: the wide GQA kernel at the reference test's shape (bs 128, 128 q heads on one kv head, 288 / 256, 4096 keys) under 1 / 2 / 3 / 4 uniform splits
and the library's choice: one piece per sequence uses half the CUs but exports no partials."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
B, Hq, Hkv, D, Dv, S, page = 128, 128, 1, 288, 256, 4096, 64
g = torch.Generator(device="cuda").manual_seed(7)
nb = B * S // page
q = torch.randn((B, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
kc = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
vc = kc[..., :Dv]
bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(B, S // page)
lens = torch.full((B,), S, dtype=torch.int32, device="cuda")
o = torch.empty((B, Hq, Dv), device="cuda", dtype=torch.bfloat16)
for n in (0, 1, 2, 3, 4, -1):
    call = lambda: torch.ops.npu.decode_gqa(q, kc, vc, o, lens, D ** -0.5, page, bt, n)
    for _ in range(100): call()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): call()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 50 * 1e3)
    print(f"splits={n}: {best:.1f} us  {B * S * D * 2 / best / 1e6:.2f} TB/s", flush=True)
