// Device-side helpers shared by the expert-parallel kernels (gfx950 / wave64 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_ep.h"

namespace mi_ep {

constexpr int kWave = 64;

struct PeerPtrs {          // W <= MI_EP_MAX_RANKS base pointers, passed by value in the kernarg segment
    void *p[MI_EP_MAX_RANKS];
};

// Device-resident call counter of a kernel family (normal dispatch, combine, low-latency dispatch).  The reference keeps its
// ping-pong / magic word in the window (cam_moe_dispatch_normal.h:273-286, notify_dispatch.h:924-938) so that a captured
// graph replays; here `ctr` points at a word of the rank's own control area holding the number of COMPLETED calls of the
// family.  Kernels launched before the call's single-workgroup exchange kernel use add = 1, that kernel stores the new value
// when it is done, kernels after it use add = 0 -- so nothing the host computes at launch (or capture) time depends on how
// many calls ran before.  ctr == nullptr: `add` is the epoch itself (explicit-epoch entry points, in-process test harness).
struct EpochRef {
    const uint64_t *ctr;
    uint64_t add;
};
__device__ __forceinline__ uint64_t epoch_of(const EpochRef &e) { return (e.ctr ? *e.ctr : 0ull) + e.add; }
// byte offset of this call's ping-pong half inside an allocation holding both halves `stride` bytes apart
struct Parity {
    EpochRef ep;
    size_t stride;
};
__device__ __forceinline__ size_t parity_off(const Parity &p) { return p.stride ? (size_t)(epoch_of(p.ep) & 1ull) * p.stride : 0; }
inline Parity make_parity(const uint64_t *ctr, uint64_t add, size_t stride) { return Parity{EpochRef{ctr, ctr ? add : 0ull}, ctr ? stride : 0}; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }

// float -> bf16 bits, round to nearest even; NaN -> 0x7FC0 (same as torch / the CPU oracle)
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f)
{
    uint32_t x = __float_as_uint(f);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
    return (x + 0x7FFFu + ((x >> 16) & 1u)) >> 16;
}

__device__ __forceinline__ float wave_max(float v)
{
    // maximum over the wave without LDS: four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
    // row_mirror), then v_permlane16_swap / v_permlane32_swap across the rows.  The shuffle form was six ds_bpermute round trips
    // (~120 cycles each), on the critical path of every token wave of the stage kernels.
#define MI_DPP_MAX(CTRL) v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false)))
    MI_DPP_MAX(0xB1);
    MI_DPP_MAX(0x4E);
    MI_DPP_MAX(0x141);
    MI_DPP_MAX(0x140);
#undef MI_DPP_MAX
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// ---- row copy by one wave.  N pieces of 1 KB (lane l moves 16 B at s[64 u], u < N): all loads back to back, then all stores, in
// straight-line code.  Written as `for u < 8: if (item < n16) v[u] = load(...)` the compiler gave every load and every store its own
// block with an s_waitcnt vmcnt(0) in front of it: sixteen serial round trips per 8 KB, each store waiting for the (possibly remote)
// acknowledgement of the one before (combine_push, pull_indexed and the low-latency pull were all compiled that way).
// NTL / NTS: nontemporal loads / stores.
template <int N, bool NTL, bool NTS>
__device__ __forceinline__ void copy_pieces(const u32x4 *s, u32x4 *d)
{
    u32x4 v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = NTL ? __builtin_nontemporal_load(s + u * kWave) : s[u * kWave];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        if (NTS) __builtin_nontemporal_store(v[u], d + u * kWave);
        else d[u * kWave] = v[u];
    }
}
// a row of n16 16-byte items: whole pieces in groups of up to eight (the group size is wave-uniform), then the items past the last
// whole piece
template <bool NTL, bool NTS>
__device__ __forceinline__ void copy_row(const u32x4 *s16, u32x4 *d16, int n16, int lane)
{
    const int nfull = n16 / kWave;
    const u32x4 *s = s16 + lane;
    u32x4 *d = d16 + lane;
    int c = 0;
    for (; c + 8 <= nfull; c += 8) copy_pieces<8, NTL, NTS>(s + c * kWave, d + c * kWave);
    switch (nfull - c) {
        case 7: copy_pieces<7, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        case 6: copy_pieces<6, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        case 5: copy_pieces<5, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        case 4: copy_pieces<4, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        case 3: copy_pieces<3, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        case 2: copy_pieces<2, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        case 1: copy_pieces<1, NTL, NTS>(s + c * kWave, d + c * kWave); break;
        default: break;
    }
    const int tail = nfull * kWave + lane;
    if (tail < n16) {
        const u32x4 v = NTL ? __builtin_nontemporal_load(s16 + tail) : s16[tail];
        if (NTS) __builtin_nontemporal_store(v, d16 + tail);
        else d16[tail] = v;
    }
}

// The same copy with WRITE-THROUGH stores (sc0 sc1: the bytes leave every cache level on their way to the destination, local or remote):
// for rows that another workgroup of the SAME launch announces to a peer -- the announcing lane then needs no release fence, only this
// wave's vmcnt drain before the workgroup counts itself in (combine_push_kernel's tail).  Stores through a buffer descriptor on the
// wave-uniform row base: the compiler counts them (raw_buffer_store aux: sc0 = 1, sc1 = 16).
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int wt_u32x4;
template <int N, bool NTL>
__device__ __forceinline__ void copy_pieces_wt(const u32x4 *s, __amdgpu_buffer_rsrc_t d, uint32_t off)
{
    u32x4 v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = NTL ? __builtin_nontemporal_load(s + u * kWave) : s[u * kWave];
#pragma unroll
    for (int u = 0; u < N; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, v[u]), d, (int)(off + (uint32_t)u * (kWave * 16u)), 0, 17);
}
template <bool NTL>
__device__ __forceinline__ void copy_row_wt(const u32x4 *s16, void *d_row /* wave-uniform */, int n16, int lane)
{
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)(uintptr_t)d_row >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)d_row);
    const __amdgpu_buffer_rsrc_t d = __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)base, 0, n16 * 16, 0x00020000);
    const int nfull = n16 / kWave;
    const u32x4 *s = s16 + lane;
    const uint32_t l16 = (uint32_t)lane * 16u;
    int c = 0;
    for (; c + 8 <= nfull; c += 8) copy_pieces_wt<8, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u));
    switch (nfull - c) {
        case 7: copy_pieces_wt<7, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        case 6: copy_pieces_wt<6, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        case 5: copy_pieces_wt<5, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        case 4: copy_pieces_wt<4, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        case 3: copy_pieces_wt<3, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        case 2: copy_pieces_wt<2, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        case 1: copy_pieces_wt<1, NTL>(s + c * kWave, d, l16 + (uint32_t)c * (kWave * 16u)); break;
        default: break;
    }
    const int tail = nfull * kWave + lane;
    if (tail < n16) {
        const u32x4 v = NTL ? __builtin_nontemporal_load(s16 + tail) : s16[tail];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, v), d, tail * 16, 0, 17);
    }
}

// Rows that a PEER announces from inside its own launch (a tag / flag word stored behind the drained write-through payload) and that a
// wave of a RUNNING launch reads right after it saw that word: the payload is read with SYSTEM-scope loads (sc0 sc1, the scope of the
// poll itself), through a buffer descriptor on the wave-uniform row base.  No cache level may answer such a load with a line it kept
// from an earlier call on the same ping-pong half (or, where neighbouring rows share a 128-byte line, from a neighbour's read), which a
// plain / nontemporal load does not promise; no acquire fence (a cache invalidate per wave) is needed either.  aux: sc0 = 1, sc1 = 16.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sys_row_rsrc(const void *row /* wave-uniform */, int bytes)
{
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)(uintptr_t)row >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)row);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 ld_sys_b128(__amdgpu_buffer_rsrc_t r, uint32_t byte_off)
{
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 17));
}
template <int N, bool NTS>
__device__ __forceinline__ void copy_pieces_sys(__amdgpu_buffer_rsrc_t s, uint32_t off, u32x4 *d)
{
    u32x4 v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = ld_sys_b128(s, off + (uint32_t)u * (kWave * 16u));
#pragma unroll
    for (int u = 0; u < N; ++u) {
        if (NTS) __builtin_nontemporal_store(v[u], d + u * kWave);
        else d[u * kWave] = v[u];
    }
}
template <bool NTS>
__device__ __forceinline__ void copy_row_sys(const void *s_row /* wave-uniform */, u32x4 *d16, int n16, int lane)
{
    const __amdgpu_buffer_rsrc_t s = sys_row_rsrc(s_row, n16 * 16);
    const int nfull = n16 / kWave;
    u32x4 *d = d16 + lane;
    const uint32_t l16 = (uint32_t)lane * 16u;
    int c = 0;
    for (; c + 8 <= nfull; c += 8) copy_pieces_sys<8, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave);
    switch (nfull - c) {
        case 7: copy_pieces_sys<7, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        case 6: copy_pieces_sys<6, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        case 5: copy_pieces_sys<5, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        case 4: copy_pieces_sys<4, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        case 3: copy_pieces_sys<3, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        case 2: copy_pieces_sys<2, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        case 1: copy_pieces_sys<1, NTS>(s, l16 + (uint32_t)c * (kWave * 16u), d + c * kWave); break;
        default: break;
    }
    const int tail = nfull * kWave + lane;
    if (tail < n16) {
        const u32x4 v = ld_sys_b128(s, (uint32_t)tail * 16u);
        if (NTS) __builtin_nontemporal_store(v, d16 + tail);
        else d16[tail] = v;
    }
}

// 8-byte words other GPUs write/poll: always system-scope atomics, never plain accesses.
__device__ __forceinline__ void sys_store_u64(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// {epoch, value} granules are their own flag (nothing else has to be published with them): relaxed.  A release per
// store would have every posting thread write the L2 back, and those write-backs serialise.
__device__ __forceinline__ void sys_store_u64_relaxed(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint64_t sys_load_u64(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Inclusive scan / sum / max over the 64 lanes of a wave on DPP (VALU only: row_shr inside the rows of 16, row_bcast:15 / :31 across
// them) instead of six ds_bpermute round trips: a scan costs ~50 cycles instead of ~600 -- the small single-workgroup kernels on the
// dispatch path (layout, notify tables, low-latency counts) are chains of such scans.  Every lane must be active.
__device__ __forceinline__ int32_t wave_incl_scan_i32(int32_t v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int32_t wave_sum_i32(int32_t v) { return __builtin_amdgcn_readlane(wave_incl_scan_i32(v), 63); }
__device__ __forceinline__ int32_t wave_max_i32(int32_t v)             // v >= 0
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

constexpr int32_t kStatusLayoutBarrier = MI_EP_STATUS_LAYOUT_BARRIER;   // the cooperative layout launch's grid barrier timed out

// error reporting word (device or pinned-host memory): first writer wins is not required, any code is enough
__device__ __forceinline__ void report_status(int32_t *status, int32_t code)
{
    __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ~100 MHz constant-rate counter (s_memrealtime); used only to bound spins.
__device__ __forceinline__ uint64_t ticks_100mhz() { return wall_clock64(); }

inline int launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MI_EP_OK : MI_EP_ELAUNCH;
}

}  // namespace mi_ep
