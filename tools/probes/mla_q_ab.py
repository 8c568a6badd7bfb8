"""Scratch: wide MLA kernel variants (8 = four KV slots, 9 = three slots + Q^T tail in LDS) at BASELINE C4, alternating in ONE process,
through the C-ABI of one or more builds of libmi_sgl_kernels.so (argv: .so paths; default the in-tree one).  Planned form, plan built once."""
import ctypes, os, sys
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
from sgl_kernel_npu.bench_hooks import _mla_inputs

libs = sys.argv[1:] or [os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_sgl_kernels.so")]
P = lambda t: c_void_p(t.data_ptr())


def setup(path):
    L = ctypes.CDLL(path)
    L.mi_mla_decode_workspace.restype = c_size_t
    L.mi_mla_decode_workspace.argtypes = [c_int, c_int, c_int]
    L.mi_mla_decode_plan_bytes.restype = c_size_t
    L.mi_mla_decode_plan_bytes.argtypes = [c_int, c_int]
    L.mi_mla_decode_build_plan.argtypes = [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]
    L.mi_mla_decode_with_plan.argtypes = [c_void_p] * 6 + [c_int] * 6 + [c_int64] * 10 + [c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
    return L


Ls = {os.path.basename(p): setup(p) for p in libs}
B, Hq, S, page = 128, 128, 4096, 64
ref = {}
for ragged in (False, True):
    q, kn, kr, bt, lens = _mla_inputs(B, Hq, S, page, ragged=ragged)
    if os.environ.get('MLA_BT') == 'linear':          # pages of a sequence consecutive in the pool (what does page placement cost?)
        bt = torch.arange(bt.numel(), dtype=torch.int32, device='cuda').reshape(bt.shape)
    out = torch.empty((B, Hq, 512), dtype=torch.bfloat16, device="cuda")
    st = c_void_p(torch.cuda.current_stream().cuda_stream)
    calls = {}
    for name, L in Ls.items():
        wsb = L.mi_mla_decode_workspace(B, Hq, -1)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        pb = L.mi_mla_decode_plan_bytes(B, 1)
        plan = torch.empty(pb // 4, dtype=torch.int32, device="cuda")
        assert L.mi_mla_decode_build_plan(P(lens), B, 1, P(plan), pb, st) == 0

        def call(L=L, ws=ws, wsb=wsb, plan=plan):
            rc = L.mi_mla_decode_with_plan(P(q), P(kn), P(kr), P(out), P(lens), P(bt), B, Hq, 1, page, bt.stride(0), S, q.stride(0), q.stride(1),
                                           kn.stride(0), kn.stride(1), kn.stride(2), kr.stride(0), kr.stride(1), kr.stride(2), out.stride(0),
                                           out.stride(1), 576 ** -0.5, 0, P(plan), P(ws), wsb, st)
            assert rc == 0, rc
        for v in (8, 9):
            calls[(name, v)] = (L, v, call)
    for _ in range(200):
        calls[next(iter(calls))][2]()
    res = {k: [] for k in calls}
    outs = {}
    for rep in range(5):
        for k, (L, v, call) in calls.items():
            L.mi_mla_decode_select_wide(v)
            for _ in range(20):
                call()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(100):
                call()
            b.record()
            torch.cuda.synchronize()
            res[k].append(a.elapsed_time(b) / 100 * 1e3)
            outs[k] = out.float().clone()
    base = outs[next(iter(calls))]
    for k in calls:
        d = (outs[k] - base).abs().max().item()
        print("ragged" if ragged else "full  ", k, "us", [round(x, 1) for x in res[k]], "maxdiff vs first", f"{d:.2e}", flush=True)
