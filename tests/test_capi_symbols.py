"""CPU-only: the C-ABI libraries load and export every symbol include/*.h declares (no compute calls)."""
import ctypes
import os

import pytest

from capi import LIBDIR, declared_symbols, load

HEADERS = {"mi_ep.h": "libmi_ep.so", "mi_sgl_kernels.h": "libmi_sgl_kernels.so"}


@pytest.mark.parametrize("header", sorted(HEADERS))
def test_library_exports_every_declared_symbol(header):
    root = os.path.dirname(LIBDIR)
    if not os.path.exists(os.path.join(os.path.dirname(root), "include", header)):
        pytest.skip(f"{header} not present yet")
    lib = load(HEADERS[header])
    names = declared_symbols(header)
    assert len(names) >= 3
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_row_geometry_helpers_are_host_callable():
    lib = load("libmi_ep.so")
    lib.mi_ep_dispatch_row_bytes.restype = ctypes.c_size_t
    lib.mi_ep_combine_row_bytes.restype = ctypes.c_size_t
    lib.mi_ep_version.restype = ctypes.c_char_p
    assert lib.mi_ep_dispatch_row_bytes(7168, 1) == 7168 + 128          # payload + 16 B of meta, rounded up to whole 128-byte lines
    assert lib.mi_ep_dispatch_row_bytes(7168, 0) == 14336 + 128
    assert lib.mi_ep_combine_row_bytes(7168) == 14336
    assert b"gfx950" in lib.mi_ep_version()


def test_notify_lds_budget_is_host_checkable():
    """The count exchange keeps W * (E + 1) counts in one workgroup's LDS; shapes beyond a CU's 160 KB are refused with MI_EP_EINVAL
    instead of failing at launch.  The helper is host-callable so a runtime can say so up front."""
    lib = load("libmi_ep.so")
    lib.mi_ep_notify_lds_bytes.restype = ctypes.c_size_t
    lib.mi_ep_notify_lds_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    f = lib.mi_ep_notify_lds_bytes
    assert f(8, 256) == (2 * 256 + 8 + 32 + 2 + 8 * 257) * 4
    assert f(8, 2048) <= 160 * 1024 and f(16, 2048) <= 160 * 1024 and f(64, 512) <= 160 * 1024      # every shape of one xGMI node, and more
    assert f(64, 2048) > 160 * 1024 and f(32, 2048) > 160 * 1024
    assert f(3, 8) == 0 and f(0, 8) == 0


def test_compact_staging_geometry_and_capacity_check():
    """mi_ep_dispatch_index_offset fixes where the expert-sorted index sits in a staging region (same value on every rank:
    it only depends on hidden, mode, top-k and the region size); stage_compact refuses a batch the region cannot hold
    before it launches anything (host-side check, callable without a GPU)."""
    import ep_harness
    lib = ep_harness.lib()          # ONE set of argtypes for the whole suite: the 16-argument form of include/mi_ep.h
    assert len(lib.mi_ep_dispatch_stage_compact.argtypes) == 16
    H, K, rb = 7168, 8, 7168 + 128          # rows start on 128-byte lines (MI_EP_ROW_STRIDE)
    region = 1 << 30
    off = lib.mi_ep_dispatch_index_offset(H, 1, K, region)
    cap = off // rb
    assert off % rb == 0 and off % 16 == 0
    assert cap * (rb + K * 8) <= region < (cap + 1) * (rb + K * 8)          # rows + index entries fill the region
    assert cap > 7 * ((1 << 30) // (K * rb))                                 # almost K times the tokens of one-row-per-(t, k) staging
    dummy = ctypes.c_void_p(0x1000)                                          # never dereferenced: the call fails its checks first
    # (x, topk_idx, idx_is_i32, send_token_idx_small, send_data_offset, T, K, H, E, rank, quant_mode, region base, region bytes,
    #  epoch counter, parity stride, stream)
    MI_EP_EINVAL = lib.mi_ep_dispatch_stage_compact(dummy, dummy, 0, dummy, dummy, cap + 1, K, H, 256, 0, 1, dummy, region, None, region, None)
    assert MI_EP_EINVAL != 0
    assert lib.mi_ep_dispatch_stage_compact(dummy, dummy, 0, dummy, dummy, 0, K, H, 256, 0, 1, dummy, region, None, region, None) == 0   # T = 0


def test_planned_decode_workspace_sizes_are_host_callable():
    """MI_MLA_SPLITS_PLANNED (-1) is passed to the workspace queries like a split count; the sizes are pure host arithmetic (256 workers when
    no device answers): partial slots for (sequences + workers + padding) items of 128 heads + flag words + the work list."""
    lib = load("libmi_sgl_kernels.so")
    for f in (lib.mi_mla_decode_workspace, lib.mi_mla_decode_plan_bytes, lib.mi_mla_decode_plan_offset, lib.mi_gqa_decode_workspace):
        f.restype = ctypes.c_size_t
    B, Hq = 128, 128
    workers = lib.mi_mla_decode_plan_workers()
    assert workers >= 1
    items = (B + workers + 7) // 8 * 8
    plan_bytes = lib.mi_mla_decode_plan_bytes(B, 1)
    assert plan_bytes == (16 + 2 * B + 4 * items) * 4
    off = lib.mi_mla_decode_plan_offset(B, Hq)
    assert off == items * 128 * 514 * 4 + B * Hq * 4
    # kv_heads is not an argument of the workspace query: the list area is sized for the most (sequence, kv head) pairs, batch * q_heads
    assert lib.mi_mla_decode_workspace(B, Hq, -1) == off + (16 + 2 * B * Hq + 4 * ((B * Hq + workers + 7) // 8 * 8)) * 4 >= off + plan_bytes
    # a 16-head shard: rows = batch * q_heads + (workers + padding) items of 16 heads
    assert lib.mi_mla_decode_plan_offset(B, 16) == (B * 16 + ((workers + 7) // 8 * 8) * 16) * 514 * 4 + B * 16 * 4
    assert lib.mi_mla_decode_workspace(B, Hq, 2) == B * Hq * 2 * 514 * 4 + B * Hq * 4          # the uniform form is unchanged
    assert lib.mi_mla_decode_workspace(0, Hq, -1) == 0 and lib.mi_mla_decode_plan_bytes(0, 1) == 0
    assert lib.mi_gqa_decode_workspace(64, 64, 128, -1) > lib.mi_gqa_decode_workspace(64, 64, 128, 4) > 0
