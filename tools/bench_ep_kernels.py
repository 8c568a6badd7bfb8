"""Scratch micro-benchmark of the EP kernels at BASELINE C2 per-rank sizes on ONE GPU (W=1 local case),
with the windows in coarse-grained (hipMalloc) vs fine-grained (hipExtMallocWithFlags) memory."""
import ctypes
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ep_harness as Hh
from capi import ptr, ptr_array, stream_ptr

hip = ctypes.CDLL("libamdhip64.so")


def fine_alloc(nbytes, flag=0x1):
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(flag))
    assert rc == 0, rc
    return p.value


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3   # us


def main():
    T, H, K, E, W = 4096, 7168, 8, 256, 1
    L_ = Hh.lib()
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
    idx = torch.topk(torch.randn((T, E), generator=g, device="cuda").abs() + 1, K, dim=-1, sorted=False)[1]
    w = torch.randn((T, K), generator=g, device="cuda")
    lay = Hh.layout(idx, E, W)
    torch.cuda.synchronize()
    R = T * K
    st = stream_ptr()
    for kind in ("coarse", "fine", "uncached"):
        rb = L_.mi_ep_dispatch_row_bytes(H, 1)
        cb = L_.mi_ep_combine_row_bytes(H)
        if kind == "coarse":
            send = torch.empty(R * rb, dtype=torch.uint8, device="cuda")
            comb = torch.empty(R * cb, dtype=torch.uint8, device="cuda")
            sp, cp = send.data_ptr(), comb.data_ptr()
        else:
            flag = 0x1 if kind == "fine" else 0x3
            sp, cp = fine_alloc(R * rb, flag), fine_alloc(R * cb, flag)
        cnt = torch.zeros((1, E + 1), dtype=torch.int32, device="cuda")
        cnt[0, :E] = lay["num_tokens_per_expert"]
        cnt[0, E] = T
        i32 = dict(dtype=torch.int32, device="cuda")
        tb = [torch.empty(E, **i32) for _ in range(9)]
        Hh.ck(L_.mi_ep_notify_tables(ptr(cnt), 1, E, 0, 0, *[ptr(t) for t in tb], None, st))
        recv_count, pull_off = tb[0], tb[8]
        recv_x = torch.empty((R, H), dtype=torch.int8, device="cuda")
        recv_s = torch.empty(R, dtype=torch.float32, device="cuda")
        src_idx = torch.empty(R * 3, **i32)
        y = torch.randn((R, H), generator=g, device="cuda").to(torch.bfloat16)
        out = torch.empty((T, H), dtype=torch.bfloat16, device="cuda")
        sp_c, cp_c = ctypes.c_void_p(sp), ctypes.c_void_p(cp)
        srcs, dsts = ptr_array([sp]), ptr_array([cp])
        t_lay = timeit(lambda: Hh.layout(idx, E, W))
        t_stage = timeit(lambda: L_.mi_ep_dispatch_stage(ptr(x), ptr(idx), 0, ptr(lay["send_token_idx_small"]),
                                                         ptr(lay["send_data_offset"]), T, K, H, E, 0, 1, sp_c, st))
        t_pull = timeit(lambda: L_.mi_ep_dispatch_pull(srcs, ptr(recv_count), ptr(pull_off), 1, E, H, 1, R, ptr(recv_x),
                                                       ptr(recv_s), ptr(src_idx), st))
        t_push = timeit(lambda: L_.mi_ep_combine_push(ptr(y), ptr(src_idx), None, R, H, K, dsts, 1, st))
        t_red = timeit(lambda: L_.mi_ep_combine_reduce(cp_c, ptr(idx), 0, ptr(w), None, None, T, K, H, E, ptr(out), st))
        gb = lambda b, us: b / us / 1e3
        print(f"[{kind}] layout {t_lay:.1f} us | stage {t_stage:.1f} us ({gb(T*H*2 + R*rb, t_stage):.0f} GB/s) | "
              f"pull {t_pull:.1f} us ({gb(2*R*rb, t_pull):.0f} GB/s) | push {t_push:.1f} us ({gb(2*R*H*2, t_push):.0f} GB/s) | "
              f"reduce {t_red:.1f} us ({gb(R*H*2 + T*H*2, t_red):.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
