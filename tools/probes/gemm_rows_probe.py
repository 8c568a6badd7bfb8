"""GEMM1 of fused_deep_moe reading its activation rows through a row-offset table (mi_ep_moe_gemm1_swiglu_rows) at the C5 shape: dense
[M, K] copy vs identity offsets vs the staged token rows of a dispatch (T rows shared by K = 8 selections each; row stride K + 16 as staged
today, and K + 128: line-aligned rows).  Outputs must be bit-identical to the dense run on the gathered copy."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, stream_ptr
L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_ep.so"))
E, H, I2, T, K = 32, 7168, 4096, 4096, 8
M = T * K
c_vp, c_int = ctypes.c_void_p, ctypes.c_int
L.mi_ep_moe_gemm1_swiglu.argtypes = [c_vp] * 5 + [c_int] * 5 + [c_vp, c_int, c_vp]
L.mi_ep_moe_gemm1_swiglu_rows.argtypes = [c_vp] * 6 + [c_int] * 5 + [c_vp, c_int, c_vp]
gen = torch.Generator().manual_seed(3)
# routing: every token picks 8 distinct experts of 32; packed order = by expert, tokens ascending
sel = torch.stack([torch.randperm(E, generator=gen)[:K] for _ in range(T)])            # [T, K]
tok = torch.arange(T)[:, None].expand(T, K).reshape(-1)
order = torch.argsort(sel.reshape(-1) * T + tok)
tok_of_row = tok[order].cuda()                                                         # [M]
cnt = torch.bincount(sel.reshape(-1), minlength=E)
cum = torch.cumsum(cnt, 0).to(torch.int32).cuda().contiguous()
asc = torch.rand(M, device="cuda")
w = torch.randint(-8, 8, (E, I2, H), dtype=torch.int8, device="cuda")
ws = torch.rand((E, I2), device="cuda")
res = {}
def timeit(f):
    for _ in range(10): assert f() == 0
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 10 * 1e3)
    return best
for stride in (H + 16, H + 128):
    staged = torch.zeros((T, stride), dtype=torch.int8, device="cuda")
    staged[:, :H] = torch.randint(-8, 8, (T, H), dtype=torch.int8, device="cuda")
    a = staged[tok_of_row, :H].contiguous()                                             # the gathered copy [M, K]
    out_d = torch.zeros((M, I2 // 2), dtype=torch.float32, device="cuda")
    out_g = torch.zeros_like(out_d)
    off_id = (torch.arange(M, device="cuda", dtype=torch.int64) * H).to(torch.int32).contiguous()        # bit pattern of uint32
    off_st = (tok_of_row.to(torch.int64) * stride).to(torch.int32).contiguous()
    t_d = timeit(lambda: L.mi_ep_moe_gemm1_swiglu(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, H, I2, ptr(out_d), 0, stream_ptr()))
    t_i = timeit(lambda: L.mi_ep_moe_gemm1_swiglu_rows(ptr(a), ptr(off_id), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, H, I2, ptr(out_g), 0, stream_ptr()))
    assert torch.equal(out_d, out_g)
    out_g.zero_()
    t_s = timeit(lambda: L.mi_ep_moe_gemm1_swiglu_rows(ptr(staged), ptr(off_st), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, H, I2, ptr(out_g), 0, stream_ptr()))
    assert torch.equal(out_d, out_g)
    print(f"row stride {stride}: dense copy {t_d:.1f} us | identity offsets {t_i:.1f} | staged token rows {t_s:.1f}   (bit-identical)", flush=True)
