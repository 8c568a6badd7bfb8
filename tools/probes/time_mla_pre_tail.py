"""Scratch: the fused GEMM2 + BMM + RoPE tail of mla_preprocess alone (128 tokens x 128 heads), event-timed; argv[1] = library."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, stream_ptr
L = ctypes.CDLL(sys.argv[1])
M, Hq = int(os.environ.get("M", 128)), 128
g = torch.Generator(device="cuda").manual_seed(1)
a8 = torch.randint(-127, 128, (M, 1536), generator=g, device="cuda", dtype=torch.int8)
wuq = torch.randint(-8, 8, (Hq * 192, 1536), generator=g, device="cuda", dtype=torch.int8)
descale = torch.rand(Hq * 192, generator=g, device="cuda") * 1e-3 + 5e-4
bias = torch.randint(-50, 50, (Hq * 192,), generator=g, device="cuda", dtype=torch.int32)
wuk_t = (torch.randn((Hq, 512, 128), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
cos, sin = torch.rand((M, 64), device="cuda").bfloat16(), torch.rand((M, 64), device="cuda").bfloat16()
q0, q1 = torch.empty((M, Hq, 512), dtype=torch.bfloat16, device="cuda"), torch.empty((M, Hq, 64), dtype=torch.bfloat16, device="cuda")
f = lambda: L.mi_mla_pre_gemm2_bmm_rope(ptr(a8), M, ptr(wuq), Hq, ptr(bias), ptr(descale), None, ptr(wuk_t), ptr(cos), ptr(sin), 0,
                                        ptr(q0), ptr(q1), None, stream_ptr())
for _ in range(200): assert f() == 0
ts = []
for _ in range(200):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); f(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print("tail us p50 %.1f min %.1f" % (ts[len(ts) // 2], ts[0]))

if hasattr(L, "mi_dbg_read_f"):
    import numpy as np
    buf = (ctypes.c_ulonglong * 256)()
    L.mi_dbg_read_f(buf)
    d = np.array(buf, dtype=np.int64).reshape(4, 64)
    for blk in range(2):
        t = d[blk]
        print("block", blk, "prologue->first wait", t[1] - t[0], "| per chunk [vmcnt wait (prev end -> wait done), barrier, work]:",
              [(int(t[1 + 3 * c] - (t[3 * c] if c else t[0])), int(t[2 + 3 * c] - t[1 + 3 * c]), int(t[3 + 3 * c] - t[2 + 3 * c])) for c in range(6)],
              "| A end -> phase B start", t[20] - t[18], "| phase B", t[21] - t[20], "| total", t[21] - t[0])

if hasattr(L, "mi_dbg_read_f_span"):      # built with -DF_SPAN: every workgroup's start / end on the 100 MHz clock
    import numpy as np
    torch.cuda.synchronize(); f(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 2048)()
    L.mi_dbg_read_f_span(buf)
    d = np.array(buf, dtype=np.int64).reshape(1024, 2)
    n = int((d[:, 1] > 0).sum())
    d = d[:n]
    t0 = d[:, 0].min()
    st, en = (d[:, 0] - t0) / 100.0, (d[:, 1] - t0) / 100.0
    du = en - st
    q = lambda a: "min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
    print("workgroups", n, "| start (us after the first):", q(st), "| duration:", q(du), "| end:", q(en))
