"""Scratch: decode_gqa at the reference test's large-group shape (128 q heads on 1 kv head, 288 / 256, batch 128 x 4096 keys): the eight-wave
LDS-DMA kernel (gqa_decode_wide.hip) against the generic one (MI_GQA_WIDE=0 in a second process), V as a view of K and as its own cache."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.attention.decode_attention import decode_gqa

g = torch.Generator(device="cuda").manual_seed(3)
B, Hq, D, Dv, S, page = 128, 128, 288, 256, 4096, 64
nb = B * S // page
q = torch.randn((B, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
kc = torch.randn((nb, page, 1, D), generator=g, device="cuda").to(torch.bfloat16)
vown = torch.randn((nb, page, 1, Dv), generator=g, device="cuda").to(torch.bfloat16)
bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(B, S // page)
o = torch.empty((B, Hq, Dv), device="cuda", dtype=torch.bfloat16)
for name, vc in (("view", kc[..., :Dv]), ("own ", vown)):
    for ragged in (False, True):
        lens = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32) if ragged else torch.full((B,), S, dtype=torch.int32, device="cuda")
        f = lambda: decode_gqa(q, kc, vc, o, lens, D ** -0.5, page, bt)
        for _ in range(50):
            f()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(50):
                f()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 50 * 1e3)
        keys = int(lens.sum().item())
        byts = keys * (D * 2 if name == "view" else (D + Dv) * 2)
        print(f"MI_GQA_WIDE={os.environ.get('MI_GQA_WIDE', '1')} V {name} {'ragged' if ragged else 'full  '}: us {[round(t, 1) for t in ts]}  "
              f"{byts / min(ts) / 1e3:.0f} GB/s algorithmic", flush=True)
