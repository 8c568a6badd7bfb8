"""decode_gqa at the reference test's shape (batch 128, 128 q heads on one kv head, 288 / 256, 4096 keys, V a view of K): pair finish off / on,
alternating in one process, full and ragged lengths.  Prints p50 per call of 200 back-to-back calls (events around each) and queued time."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
import sgl_kernel_npu.attention.decode_attention  # noqa
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "lib", "libmi_sgl_kernels.so"))
B, Hq, D, Dv, S, page = 128, 128, 288, 256, 4096, 64
maxp = S // page
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn((B, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
k = torch.randn((B * maxp, page, 1, D), generator=g, device="cuda").to(torch.bfloat16)
v = k[..., :Dv]
bt = torch.randperm(B * maxp, device="cuda").to(torch.int32).reshape(B, maxp)
out = torch.empty((B, Hq, Dv), dtype=torch.bfloat16, device="cuda")
full = torch.full((B,), S, dtype=torch.int32, device="cuda")
rag = torch.randint(1, S + 1, (B,), generator=g, device="cuda").to(torch.int32)

def run(lens, n=200):
    for _ in range(50):
        torch.ops.npu.decode_gqa(q, k, v, out, lens, D ** -0.5, page, bt, 0)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); torch.ops.npu.decode_gqa(q, k, v, out, lens, D ** -0.5, page, bt, 0); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        torch.ops.npu.decode_gqa(q, k, v, out, lens, D ** -0.5, page, bt, 0)
    b.record(); torch.cuda.synchronize()
    return t[n // 2], a.elapsed_time(b) * 1e3 / n

for rep in range(3):
    for mode in (0, 1):
        L.mi_gqa_decode_set_pair(mode)
        f = run(full); r = run(rag)
        print(f"rep {rep} pair={mode}: full p50 {f[0]:.1f} us queued {f[1]:.1f} | ragged p50 {r[0]:.1f} queued {r[1]:.1f}", flush=True)
L.mi_gqa_decode_set_pair(-1)
