"""Scratch: stage / barrier stamps of the one-launch mla_preprocess (library built with -DMEGA_TIMING, LD_PRELOADed in front of the product one):
    tools/build_timing.sh mega -DMEGA_TIMING && LD_PRELOAD=sgl-kernel-npu_amd/lib/timing/libmi_sgl_kernels_mega.so python tools/probes/time_mla_pre_mega.py"""
import ctypes, os, runpy, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("MLA_PRE_QUANT", "per_tensor_quant_asymm")
runpy.run_path(os.path.join(HERE, "time_mla_pre_op.py"), run_name="__main__")
L = ctypes.CDLL(os.environ["LD_PRELOAD"].split(":")[0])
buf = (ctypes.c_ulonglong * (1024 * 8))()
assert L.mi_dbg_read_mega(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
a = (a - t0) / 100.0
names = ["start", "stage0 quant done", "barrier0 passed", "stage1 GEMM1 done", "barrier1 passed", "stage2 middle done", "barrier2 passed", "end"]
print(len(a), "workgroups; us since the first start: [min, p50, max]")
for i, n in enumerate(names):
    print("  %-20s %6.2f %6.2f %6.2f" % (n, a[:, i].min(), np.percentile(a[:, i], 50), a[:, i].max()))
