"""Scratch: phase timestamps (100 MHz) of notify_exchange_tables at W = 1 (library built with -DNOTIFY_TIMING)."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from capi import ptr, ptr_array, stream_ptr
from ctypes import c_int, c_size_t, c_uint32, c_uint64, c_void_p
L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "timing_ep", "libmi_ep_t.so"))
V, I = c_void_p, c_int
L.mi_ep_notify_exchange_tables.argtypes = [V, V, V, I, V, c_uint32, V, c_uint64, V, I, I, I, I] + [V] * 10 + [V, c_size_t, V, I, V, V]
W, E = 1, 256
i32 = dict(dtype=torch.int32, device="cuda")
notify = torch.zeros(2 * W * (E + 1) + 64, dtype=torch.int64, device="cuda")
flags = torch.zeros(64, dtype=torch.int64, device="cuda")
cnt_in = torch.full((E,), 128, **i32)
cnt = torch.zeros(W * (E + 1) + 64, **i32)
tabs = [torch.zeros(max(E, 1), **i32) for _ in range(9)]
ctr = torch.zeros(1, dtype=torch.int64, device="cuda")
status = torch.zeros(4, **i32)
summ = torch.zeros(4 + 2048, **i32)
for it in range(5):
    rc = L.mi_ep_notify_exchange_tables(ptr_array([notify.data_ptr()]), ptr_array([flags.data_ptr()]), ptr(cnt_in), 4096, ptr(notify), 0, ptr(flags), 0,
                                        ptr(cnt), W, E, 0, 0, *[ptr(t) for t in tabs], ptr(summ), ptr(ctr), (W * (E + 1) + 32) * 8, ptr(status), 2000, None, stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
o = (W * (E + 1) + 17) & ~1
tk = cnt[o: o + 12].view(torch.int64).cpu().tolist()
d = [(tk[i + 1] - tk[i]) / 100 for i in range(4)]
print("us [epoch read + post, granule + flag wait, fence + barrier, tables + summary] =", [round(v, 2) for v in d], "total", round(sum(d), 2), "status", status.tolist())
