"""Scratch: split_qkv_rmsnorm_rope / _mrope / _rope_pos_cache at 4096 x 8192, 200 queued calls between two events (kernel time + launch gap)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu  # noqa: F401
from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope import split_qkv_rmsnorm_rope
from sgl_kernel_npu.norm.split_qkv_rmsnorm_mrope import triton_split_qkv_rmsnorm_mrope
from sgl_kernel_npu.norm.split_qkv_rmsnorm_rope_pos_cache_half_npu import split_qkv_rmsnorm_rope_pos_cache_half_npu
B = 4096
g = torch.Generator(device="cuda").manual_seed(0)
xt = torch.randn((B, 8192), generator=g, device="cuda").to(torch.bfloat16)
sn, cs = torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16), torch.rand((B, 1, 1, 128), device="cuda").to(torch.bfloat16)
hw = torch.randn(128, device="cuda").to(torch.bfloat16)
cs3 = torch.rand((3, B, 128), device="cuda").to(torch.bfloat16)
cache = torch.randn((8192, 128), device="cuda")
posb = torch.randint(0, 8192, (B,), device="cuda")
forms = {
    "rope": lambda: split_qkv_rmsnorm_rope(xt, sn, cs, 6144, 1024, 128, 1e-6, hw, hw, hw, hw),
    "mrope interleaved": lambda: triton_split_qkv_rmsnorm_mrope(xt, hw, hw, cs3, 48, 8, 128, 1e-6, [24, 20, 20], True),
    "mrope contiguous": lambda: triton_split_qkv_rmsnorm_mrope(xt, hw, hw, cs3, 48, 8, 128, 1e-6, [24, 20, 20], False),
    "pos cache": lambda: split_qkv_rmsnorm_rope_pos_cache_half_npu(xt, posb, cache, 6144, 1024, 128, eps=1e-6, q_weight=hw, k_weight=hw),
}
for name, f in forms.items():
    for _ in range(50): f()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200): f()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 200 * 1e3)
    print(f"{name}: {best:.1f} us  {B * 8192 * 4 / best / 1e6:.2f} TB/s", flush=True)
# fused_rope_qk_mqa (A13): MLA-sized q [T, 128, 192] with the first 64 dims rotated, one shared key head; and a pure-rope shape [T, 128, 64]
from sgl_kernel_npu.norm.fused_rope_qk_mqa import fused_rope_qk_mqa
for (T, Hq, Hk, D, R) in ((4096, 128, 1, 192, 64), (4096, 128, 1, 64, 64), (128, 128, 1, 192, 64)):
    qq = torch.randn((T, Hq, D), device="cuda").to(torch.bfloat16)
    kk = torch.randn((T, Hk, D), device="cuda").to(torch.bfloat16)
    cs = torch.rand((T, R), device="cuda").to(torch.bfloat16)
    for neox in (True, False):
        f = lambda: fused_rope_qk_mqa(qq, kk, cs, R, neox)
        for _ in range(20): f()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100): f()
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 100 * 1e3)
        byts = T * (Hq + Hk) * D * 4
        print(f"fused_rope_qk_mqa T={T} Hq={Hq} D={D} R={R} neox={neox}: {best:.1f} us  {byts / best / 1e6:.2f} TB/s", flush=True)
