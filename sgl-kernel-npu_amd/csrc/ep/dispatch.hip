// A3 normal dispatch for gfx950: stage (quantise + pack into the own send window) and pull
// (receiver gathers its segments from every source's window straight into the output tensors).
// Replaces aclnnCamMoeDispatchNormal (reference kernel csrc/deepep/ops/op_kernel/cam_moe_dispatch_normal.h:
// QuantProcess :326-363, FillTriple :366-375, InputToShare :440-473, ShareToOutputLongSeq :717-760).
//
// MI355X design
//  * one wave64 per token: the 14 KB bf16 row is read once with 16-B loads (all of it in flight at
//    once, 56 VGPRs), |max| is a 6-step wave shuffle (no LDS, no barrier), the row is quantised once and
//    the K copies are written as 16-B/lane stores (1 KiB per wave-instruction, fully coalesced) --
//    the reference re-reads and re-quantises the row for every k;
//  * every staged row carries its 16-byte meta {scale, token, k, src_rank} so one contiguous pull
//    moves payload + scale + triple; rows are 16-B aligned (H % 16 == 0);
//  * pull = one wave per output row, segment lookup by binary search in an LDS copy of recv_count;
//    consecutive workgroups walk the (local expert, src rank) order, so all 7 xGMI links carry reads
//    at the same time; 8 x 16 B per lane in flight.
// HBM roofline (per rank, algorithmic): stage reads T*H*2 and writes n_pairs*(H+16); pull reads and
// writes n_recv*(H+16).
#include "device_once.h"
#include <stdlib.h>

#include <algorithm>

#include "ep_common.h"
#include "layout_dev.h"

namespace mi_ep {

constexpr int kStageWaves = 4;      // tokens per workgroup
constexpr int kMaxItems = MI_EP_MAX_HIDDEN / 16 / kWave;   // 16-element items per lane (8)

template <bool I32>
__device__ __forceinline__ long long ld_idx(const void *p, long long i)
{
    if (I32) return (long long)((const int32_t *)p)[i];
    return ((const long long *)p)[i];
}

// Where a (t,k) row goes.  Normal mode (L == 0): own send window (dsts.p[0]), slot = send_data_offset[e] +
// send_token_idx_small (cam_moe_dispatch_normal.h:445-448).  Low-latency mode: the destination rank's window,
// region (le, my_rank), position send_token_idx_small (moe_distribute_dispatch_v2.h:668-672).
struct LLGeom {
    int L, W, max_tokens;
};
// Push transport of normal dispatch (mi_ep_dispatch_stage_push): the quantised token row is written ONCE into the window of
// every rank that owns at least one of its experts -- slab `my_rank` of that rank's dispatch region, row t -- and the
// expert-sorted index entries {t, k} follow it into the same slab.  L == 0: not a push (compact staging in the own window).
struct PushGeom {
    int L;             // local experts per rank
    size_t slab;       // bytes of one source slab = region_bytes / W (rounded down to 256)
};
// compact / push staging of one token: `write_row(base)` stores the payload + meta at row t of the slab at `base`
template <class WriteRow>
__device__ __forceinline__ void stage_token_rows(const PushGeom &pg, const PeerPtrs &dsts, size_t poff, size_t idx_off, int my_rank,
                                                 int t, int K, long long e_l, int slot_l, const int32_t *send_off, WriteRow write_row)
{
    const int lane = lane_id();
    if (pg.L == 0) {
        uint8_t *base = (uint8_t *)dsts.p[0] + poff;
        write_row(base);
        if (lane < K && e_l >= 0) ((uint2 *)(base + idx_off))[slot_l] = uint2{(uint32_t)t, (uint32_t)lane};
        return;
    }
    // distinct destination ranks of this token (K <= 16 selections, W <= 64 ranks): one row per rank, one index entry per pair
    const int d_l = e_l >= 0 ? (int)((uint32_t)e_l / (uint32_t)pg.L) : -1;      // 0 <= e_l < E: 32-bit division (the 64-bit one is ~150 VALU operations)
    unsigned long long rmask = 0ull;
    for (int k = 0; k < K; ++k) {
        const int dk = __builtin_amdgcn_readlane(d_l, k);                   // k is wave-uniform: v_readlane, not ds_bpermute
        if (dk >= 0) rmask |= 1ull << dk;
    }
    while (rmask) {                                   // wave-uniform
        const int d = __builtin_ctzll(rmask);
        rmask &= rmask - 1;
        uint8_t *base = (uint8_t *)dsts.p[d] + poff + (size_t)my_rank * pg.slab;
        write_row(base);
        // position among the rows this rank sends to d, in its expert-sorted order (= the receiver's relative pull offset)
        if (d_l == d) ((uint2 *)(base + idx_off))[slot_l - send_off[d * pg.L]] = uint2{(uint32_t)t, (uint32_t)lane};
    }
}
__device__ __forceinline__ void route(const LLGeom &ll, int e, int small, const int32_t *send_off, int my_rank,
                                      int &slot, int &dst)
{
    if (ll.L == 0) {
        slot = send_off[e] + small;
        dst = 0;
    } else {
        dst = e / ll.L;
        slot = ((e % ll.L) * ll.W + my_rank) * ll.max_tokens + small;
    }
}

// Self-routing (low-latency dispatch in one launch, ll_layout_send_kernel): a send wave finds the slab position of ITS (token, selection)
// itself -- the number of earlier pairs, in row-major (t, k) order, that selected the same expert, which is what the layout's
// send_token_idx_small holds -- from a copy of the batch's routing table in LDS (<= 8192 ids; every workgroup loads it once, one id
// per thread), 64 pairs per step: compare, ballot, popcount.  No wave waits for the layout workgroup; the row is loaded and quantised
// first, so the count runs under the row's memory latency.
// buffer descriptor over one window row (wave-uniform base): the write-through stores of the one-launch low-latency form go through it
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(void *row, int bytes)
{
    const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)(uintptr_t)row >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)row);
    return __builtin_amdgcn_make_buffer_rsrc((void *)(uintptr_t)base, 0, bytes, 0x00020000);
}
struct LateIdx {
    const int32_t *ids;        // LDS: expert id of pair p = t * K + k, -1 = no selection; NULL: the layout ran in an earlier launch
    int kpart;                 // the selection this wave sends
    uint32_t tag;              // != 0 (tagged rows, mi_ep_ll_dispatch_layout_send_tagged): the row's meta word 3 = src_rank | tag << 8, written
                               // only when the row's payload has drained -- the receiver waits for the tag instead of for a count exchange
};
// 24-bit, never zero: the tag of the call with epoch `ep64` (a slab half is rewritten every second call, so a stale row carries another tag)
__device__ __forceinline__ uint32_t ll_row_tag(uint64_t ep64) { return (uint32_t)(ep64 % 0xFFFFFFull) + 1u; }
template <bool I32, bool LATE>
__device__ __forceinline__ void route_token(const LLGeom &ll, const void *topk_idx, const int32_t *idx_small, const int32_t *send_off, int t,
                                            int K, int E, int my_rank, const LateIdx &late, long long &e_l, int &slot_l, int &dst_l)
{
    const int lane = lane_id();
    if (LATE) {
        const int p = t * K + late.kpart;                        // wave-uniform
        const int e = late.ids[p];
        if (e < 0) return;
        int before = 0;
        for (int c = 0; c < p; c += kWave) {
            const int i = c + lane;
            before += __popcll(__ballot(i < p && late.ids[i] == e));
        }
        if (lane == late.kpart) route(ll, e, before, send_off, my_rank, slot_l, dst_l);
        return;
    }
    if (lane < K && e_l >= 0) route(ll, (int)e_l, idx_small[(long long)t * K + lane], send_off, my_rank, slot_l, dst_l);
}

// ---------------------------------------------------------------------------------------------
// stage, INT8: item = 16 consecutive elements = two 16-B loads -> one 16-B store
// ---------------------------------------------------------------------------------------------
// QM = MI_EP_QUANT_INT8 / MI_EP_QUANT_INT8_NOEPS / MI_EP_QUANT_FP8_E4M3 (one byte per element each, same row layout)
template <bool I32, int QM, bool LATE>
__device__ __forceinline__ void stage_int8_body(
    const uint16_t *__restrict__ x, const void *__restrict__ topk_idx, const int32_t *__restrict__ idx_small,
    const int32_t *__restrict__ send_off, int T, int K, int H, int E, int my_rank, const PeerPtrs &dsts, const LLGeom &ll, int ksplit,
    size_t idx_off, const PushGeom &pg, const Parity &par, const int wid, const LateIdx &late)
{
    const int lane = lane_id();
    // ksplit waves share a token (decode-size batches: every wave re-reads the row from L2 and writes K / ksplit copies)
    const int t = wid / ksplit, kpart = wid - t * ksplit;
    if (t >= T) return;
    const int nitems = H / 16;
    const size_t stride = MI_EP_ROW_STRIDE(H);
    // the row is requested before the routing is known (a token that selects nothing is the rare case): behind the routing's two
    // dependent loads (expert id, then slot and segment offset) the 14 KB of the row started one to two round trips late
    const u32x4 *src = (const u32x4 *)(x + (size_t)t * H);
    u32x4 raw[kMaxItems][2];
    float amax = 0.f;
#pragma unroll
    for (int it = 0; it < kMaxItems; ++it) {
        const int item = it * kWave + lane;
        if (item < nitems) {
            raw[it][0] = src[item * 2];
            raw[it][1] = src[item * 2 + 1];
        } else {
            raw[it][0] = raw[it][1] = u32x4{0, 0, 0, 0};
        }
    }
    // routing of this token: lane k < K owns pair (t,k)
    long long e_l = -1;
    int slot_l = 0, dst_l = 0;
    if (lane < K) {
        e_l = ld_idx<I32>(topk_idx, (long long)t * K + lane);
        if (e_l < 0 || e_l >= E) e_l = -1;
    }
    if (!LATE) route_token<I32, false>(ll, topk_idx, idx_small, send_off, t, K, E, my_rank, late, e_l, slot_l, dst_l);
    const unsigned long long vmask = __ballot(e_l >= 0);
    if (vmask == 0ull) return;      // token selects nothing: no row is produced
    // |x| of a bf16 orders like its bit pattern as an unsigned 16-bit integer, so the row maximum is taken on the packed words
    // (v_pk_max_u16, two elements per instruction, no unpacking): the SIMDs are 16 lanes wide, every wave instruction costs 4 cycles,
    // and this kernel spent as long in its ~1260 VALU instructions per token as in its memory phases
    {
        typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
        u16x2 m2 = u16x2{0, 0};
#pragma unroll
        for (int it = 0; it < kMaxItems; ++it)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, raw[it][h][j] & 0x7FFF7FFFu));
        const uint32_t m = m2[0] > m2[1] ? m2[0] : m2[1];
        // a NaN element (pattern above 0x7F80) wins the integer maximum: such a row gets amax = inf (its bytes are meaningless either way)
        amax = __uint_as_float((m > 0x7F80u ? 0x7F80u : m) << 16);
    }
    amax = wave_max(amax);
    float s, scale_out;
    if (QM == MI_EP_QUANT_INT8) {
        s = 127.0f / (amax + 1e-12f);
        scale_out = 1.0f / s;
    } else if (QM == MI_EP_QUANT_INT8_NOEPS) {
        s = (amax == 0.f) ? 0.f : 127.0f / amax;       // all-zero row: q = 0, scale = 0 (see oracle)
        scale_out = (amax == 0.f) ? 0.f : 1.0f / s;
    } else {                                           // per-token FP8 E4M3 (moe_distribute_dispatch_v2_a5.h:1130-1131,1154-1155)
        s = amax > 0.f ? 448.0f / amax : 1.0f;
        scale_out = 1.0f / s;
    }
    u32x4 q[kMaxItems];
    // the raw row is made opaque between the two passes: otherwise the compiler keeps the 128 floats it unpacked for the maximum alive
    // for the quantisation (161 VGPRs, 3 waves per SIMD -- the 4096 token waves of a C2 batch then need two rounds); unpacking again
    // costs one shift / mask per element
#pragma unroll
    for (int it = 0; it < kMaxItems; ++it) asm volatile("" : "+v"(raw[it][0]), "+v"(raw[it][1]));
#pragma unroll
    for (int it = 0; it < kMaxItems; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 s2 = f32x2{s, s};
                const uint32_t w0 = raw[it][h][jj * 2], w1 = raw[it][h][jj * 2 + 1];
                uint32_t p0, p1 = 0u;
                if (QM == MI_EP_QUANT_FP8_E4M3) {
                    // fp32 product, then the hardware's round-to-nearest-even conversion to OCP E4M3 (two elements per instruction,
                    // the 16-bit half of the destination selected by the last operand); |x * s| <= 448 by construction
                    const f32x2 r0 = f32x2{bf16_to_f32(w0 & 0xFFFFu), __uint_as_float(w0 & 0xFFFF0000u)} * s2;
                    const f32x2 r1 = f32x2{bf16_to_f32(w1 & 0xFFFFu), __uint_as_float(w1 & 0xFFFF0000u)} * s2;
                    p0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(r0[0], r0[1], 0, false);
                    p0 = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(r1[0], r1[1], (int)p0, true);
                } else {
                // round(x * s) through the 1.5 * 2^23 trick: the float sum's low mantissa bits are the nearest-even integer in two's
                // complement (|x * s| <= 127), i.e. rintf + cvt in one addition, on two elements per instruction (v_pk_mul_f32 /
                // v_pk_add_f32); the four low bytes are gathered with v_perm_b32
                const f32x2 magic = f32x2{12582912.0f, 12582912.0f};
                const f32x2 r0 = f32x2{bf16_to_f32(w0 & 0xFFFFu), __uint_as_float(w0 & 0xFFFF0000u)} * s2 + magic;
                const f32x2 r1 = f32x2{bf16_to_f32(w1 & 0xFFFFu), __uint_as_float(w1 & 0xFFFF0000u)} * s2 + magic;
                p0 = __builtin_amdgcn_perm(__float_as_uint(r0[1]), __float_as_uint(r0[0]), 0x0c0c0400u);   // bytes: lo, hi, 0, 0
                p1 = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x04000c0cu);   // bytes: 0, 0, lo, hi
                }
                q[it][h * 2 + jj] = p0 | p1;
            }
    }
    if (LATE) route_token<I32, true>(ll, topk_idx, idx_small, send_off, t, K, E, my_rank, late, e_l, slot_l, dst_l);
    if (idx_off) {
        // compact staging (normal-mode pull transport): the row is written ONCE at slot t; the expert-sorted index tells the
        // receivers which token row each of their rows is (K-fold less staging traffic than one copy per (t, k)).
        // Push transport: the same row + index entries, written into every destination rank's window instead of the own one.
        stage_token_rows(pg, dsts, parity_off(par), idx_off, my_rank, t, K, e_l, slot_l, send_off, [&](uint8_t *base) {
            uint8_t *row = base + (size_t)t * stride;
            u32x4 *dst = (u32x4 *)row;
#pragma unroll
            for (int it = 0; it < kMaxItems; ++it) {
                const int item = it * kWave + lane;
                if (item < nitems) dst[item] = q[it];
            }
            if (lane == 0) *(u32x4 *)(row + H) = u32x4{__float_as_uint(scale_out), (uint32_t)t, 0u, (uint32_t)my_rank};
        });
        return;
    }
    const size_t poff = parity_off(par);
    for (int k = kpart; k < K; k += ksplit) {
        if (!((vmask >> k) & 1ull)) continue;          // wave-uniform
        const int slot = __builtin_amdgcn_readlane(slot_l, k);      // k is wave-uniform: v_readlane instead of an LDS round trip
        const int drank = __builtin_amdgcn_readlane(dst_l, k);
        uint8_t *row = (uint8_t *)dsts.p[drank] + poff + (size_t)slot * stride;
        const u32x4 meta = u32x4{__float_as_uint(scale_out), (uint32_t)t, (uint32_t)k, (uint32_t)my_rank};
        if (LATE) {
            // one-launch low-latency form: the rows are announced to their owners from INSIDE this launch (ll_layout_send_kernel's tail), so
            // they are written through the caches (sc0 sc1) -- the announcing workgroup then needs no release fence, only every wave's drain
            const __amdgpu_buffer_rsrc_t d = row_rsrc(row, H + MI_EP_ROW_META_BYTES);
#pragma unroll
            for (int it = 0; it < kMaxItems; ++it) {
                const int item = it * kWave + lane;
                if (item < nitems) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, q[it]), d, item * 16, 0, 17);
            }
            u32x4 m = meta;
            if (late.tag) {                                 // tagged rows: the payload is at its owner before the meta word says so
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                m[3] = (uint32_t)my_rank | (late.tag << 8);
            }
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, m), d, H, 0, 17);
            continue;
        }
        u32x4 *dst = (u32x4 *)row;
#pragma unroll
        for (int it = 0; it < kMaxItems; ++it) {
            const int item = it * kWave + lane;
            if (item < nitems) dst[item] = q[it];
        }
        if (lane == 0) *(u32x4 *)(row + H) = meta;
    }
}

template <bool I32, int QM>
__global__ __launch_bounds__(kWave * kStageWaves) void stage_int8_kernel(
    const uint16_t *__restrict__ x, const void *__restrict__ topk_idx, const int32_t *__restrict__ idx_small,
    const int32_t *__restrict__ send_off, int T, int K, int H, int E, int my_rank, PeerPtrs dsts, LLGeom ll, int ksplit, size_t idx_off,
    PushGeom pg, Parity par)
{
    stage_int8_body<I32, QM, false>(x, topk_idx, idx_small, send_off, T, K, H, E, my_rank, dsts, ll, ksplit, idx_off, pg, par,
                                    (int)(blockIdx.x * kStageWaves + threadIdx.x / kWave), LateIdx{nullptr, 0, 0u});
}

// stage, BF16 (no quantisation): item = one 16-B chunk
template <bool I32, bool LATE>
__device__ __forceinline__ void stage_bf16_body(
    const uint16_t *__restrict__ x, const void *__restrict__ topk_idx, const int32_t *__restrict__ idx_small,
    const int32_t *__restrict__ send_off, int T, int K, int H, int E, int my_rank, const PeerPtrs &dsts, const LLGeom &ll, int ksplit,
    size_t idx_off, const PushGeom &pg, const Parity &par, const int wid, const LateIdx &late)
{
    const int lane = lane_id();
    // ksplit waves share a token (decode-size batches: every wave re-reads the row from L2 and writes K / ksplit copies)
    const int t = wid / ksplit, kpart = wid - t * ksplit;
    if (t >= T) return;
    const int nitems = H / 8;
    const size_t stride = MI_EP_ROW_STRIDE((size_t)H * 2);
    long long e_l = -1;
    int slot_l = 0, dst_l = 0;
    if (lane < K) {
        e_l = ld_idx<I32>(topk_idx, (long long)t * K + lane);
        if (e_l < 0 || e_l >= E) e_l = -1;
    }
    if (!LATE) route_token<I32, false>(ll, topk_idx, idx_small, send_off, t, K, E, my_rank, late, e_l, slot_l, dst_l);
    const unsigned long long vmask = __ballot(e_l >= 0);
    if (vmask == 0ull) return;
    const u32x4 *src = (const u32x4 *)(x + (size_t)t * H);
    constexpr int kIt = kMaxItems * 2;
    u32x4 raw[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int item = it * kWave + lane;
        if (item < nitems) raw[it] = src[item];
    }
    if (LATE) route_token<I32, true>(ll, topk_idx, idx_small, send_off, t, K, E, my_rank, late, e_l, slot_l, dst_l);
    if (idx_off) {                                  // compact / push staging, see stage_int8_kernel
        stage_token_rows(pg, dsts, parity_off(par), idx_off, my_rank, t, K, e_l, slot_l, send_off, [&](uint8_t *base) {
            uint8_t *row = base + (size_t)t * stride;
            u32x4 *dst = (u32x4 *)row;
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int item = it * kWave + lane;
                if (item < nitems) dst[item] = raw[it];
            }
            if (lane == 0) *(u32x4 *)(row + (size_t)H * 2) = u32x4{0u, (uint32_t)t, 0u, (uint32_t)my_rank};
        });
        return;
    }
    const size_t poff = parity_off(par);
    for (int k = kpart; k < K; k += ksplit) {
        if (!((vmask >> k) & 1ull)) continue;
        const int slot = __builtin_amdgcn_readlane(slot_l, k);      // k is wave-uniform: v_readlane instead of an LDS round trip
        const int drank = __builtin_amdgcn_readlane(dst_l, k);
        uint8_t *row = (uint8_t *)dsts.p[drank] + poff + (size_t)slot * stride;
        const u32x4 meta = u32x4{0u, (uint32_t)t, (uint32_t)k, (uint32_t)my_rank};
        if (LATE) {                                     // written through the caches: see stage_int8_body
            const __amdgpu_buffer_rsrc_t d = row_rsrc(row, H * 2 + MI_EP_ROW_META_BYTES);
#pragma unroll
            for (int it = 0; it < kIt; ++it) {
                const int item = it * kWave + lane;
                if (item < nitems) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, raw[it]), d, item * 16, 0, 17);
            }
            u32x4 m = meta;
            if (late.tag) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                m[3] = (uint32_t)my_rank | (late.tag << 8);
            }
            if (lane == 0) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, m), d, H * 2, 0, 17);
            continue;
        }
        u32x4 *dst = (u32x4 *)row;
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int item = it * kWave + lane;
            if (item < nitems) dst[item] = raw[it];
        }
        if (lane == 0) *(u32x4 *)(row + (size_t)H * 2) = meta;
    }
}
template <bool I32>
__global__ __launch_bounds__(kWave * kStageWaves) void stage_bf16_kernel(
    const uint16_t *__restrict__ x, const void *__restrict__ topk_idx, const int32_t *__restrict__ idx_small,
    const int32_t *__restrict__ send_off, int T, int K, int H, int E, int my_rank, PeerPtrs dsts, LLGeom ll, int ksplit, size_t idx_off,
    PushGeom pg, Parity par)
{
    stage_bf16_body<I32, false>(x, topk_idx, idx_small, send_off, T, K, H, E, my_rank, dsts, ll, ksplit, idx_off, pg, par,
                                (int)(blockIdx.x * kStageWaves + threadIdx.x / kWave), LateIdx{nullptr, 0, 0u});
}

// The count exchange of a low-latency dispatch, run by ONE workgroup (of any size up to 1024 threads): post this rank's per-expert counts
// to every peer (optional), wait (bounded) for the L*W count granules of this call's epoch, inclusive cumsum in idx-i order, per-expert
// counts, and complete the family's call counter.  c = LDS, L*W + 16 words.
__device__ __forceinline__ void ll_counts_body(int32_t *c, const PeerPtrs &count_peers, const int32_t *__restrict__ my_counts_out /*[E] or null*/,
                                               int my_rank, const uint64_t *__restrict__ granules_base, uint64_t ep64,
                                               size_t counts_parity_stride, uint64_t *epoch_bump, int L, int W, int count_type,
                                               int32_t *__restrict__ layout_range, int64_t *__restrict__ packed_recv_count, int32_t *status,
                                               uint64_t timeout_ticks)
{
    const int LW = L * W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t *wave_tot = c + LW;
    // this call's epoch (device-resident counter + 1, or the explicit value) and the ping-pong half of the count granules
    const uint32_t epoch = (uint32_t)ep64;
    const size_t cpoff = (size_t)(ep64 & 1ull) * counts_parity_stride;
    const uint64_t *granules = (const uint64_t *)((const uint8_t *)granules_base + cpoff);
    // optional fused post (one rank per process): this rank's per-expert counts to every peer, then collect everybody's
    if (my_counts_out) {
        for (int i = tid; i < LW; i += blockDim.x) {
            const int d = i / L, le = i % L;
            sys_store_u64_relaxed((uint64_t *)((uint8_t *)count_peers.p[d] + cpoff) + (size_t)le * W + my_rank, ((uint64_t)epoch << 32) | (uint32_t)my_counts_out[d * L + le]);
        }
    }
    const uint64_t t0 = ticks_100mhz();
    for (int i = tid; i < LW; i += blockDim.x) {
        uint64_t g;
        while (((g = sys_load_u64(granules + i)) >> 32) != epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                report_status(status, 2000 + i);
                g = 0;
                break;
            }
        }
        c[i] = (int32_t)(uint32_t)g;
    }
    __syncthreads();
    // inclusive scan of c[0..LW): each thread owns a contiguous chunk, wave scan of the chunk sums, then wave offsets
    const int per = (LW + blockDim.x - 1) / blockDim.x;
    const int b0 = min(LW, tid * per), b1 = min(LW, b0 + per);
    int32_t sum = 0;
    for (int i = b0; i < b1; ++i) sum += c[i];
    const int32_t inc = wave_incl_scan_i32(sum);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int32_t run = inc - sum;
    for (int w = 0; w < wave; ++w) run += wave_tot[w];
    for (int i = b0; i < b1; ++i) {
        run += c[i];
        c[i] = run;
    }
    __syncthreads();
    for (int i = tid; i < LW; i += blockDim.x) layout_range[i] = c[i];
    for (int le = tid; le < L; le += blockDim.x) {
        const int32_t end = c[(le + 1) * W - 1], beg = le ? c[le * W - 1] : 0;
        packed_recv_count[le] = (count_type == 0) ? (int64_t)end : (int64_t)(end - beg);
    }
    // the call is now "complete" as far as later kernels of the family are concerned: they read the counter with add = 0
    if (epoch_bump && tid == 0) *epoch_bump = ep64;
}

// The count exchange raised from INSIDE the layout + send launch (ll_layout_send_kernel with a tail): every workgroup -- the send
// workgroups after their write-through rows have drained, the layout workgroup after an agent-scope release of its tables -- counts
// itself in at a device word of the rank's control area; the last one to arrive runs ll_counts_body and re-arms the word.  One launch
// and one kernel boundary less per low-latency dispatch (three launches -> two).
struct LLTail {
    uint32_t *arrive;             // NULL: no tail (the count exchange is a launch of its own)
    PeerPtrs count_peers;
    const uint64_t *granules_base;
    size_t counts_parity_stride;
    uint64_t *epoch_bump;         // the family's completed-call counter (this call's epoch = counter + 1)
    int L, count_type;
    int32_t *layout_range;
    int64_t *packed_recv_count;
    int32_t *status;
    uint64_t timeout_ticks;
    // TAGGED form (arrive == NULL, cur_epoch set; mi_ep_ll_dispatch_layout_send_tagged): nobody waits for anybody in this launch -- the layout
    // workgroup posts this rank's per-expert counts to the peers as soon as it has them and leaves the call's epoch at *cur_epoch, the send
    // waves tag their rows; the packing launch (ll_wait_pack_kernel) collects counts and rows itself.
    uint64_t *cur_epoch;
};

// Low-latency dispatch, layout + send in ONE launch of 1024-thread workgroups: workgroup 0 computes the layout tables of the batch
// (<= 1024 tokens: one workgroup of layout_small_body; the count exchange that follows needs num_tokens_per_expert, the handle the rest);
// workgroups 1.. are the send waves, one per (token, selection), which route themselves from an LDS copy of the routing table (above).
// No workgroup waits for any other.  Same rows, tables and bytes as mi_ep_dispatch_layout + mi_ep_ll_dispatch_send.
constexpr int kLLSendWaves = 4;
template <bool I32, int QM, int UT>
__global__ __launch_bounds__(1024) void ll_layout_send_kernel(
    const uint16_t *__restrict__ x, const void *__restrict__ topk_idx, int T, int K, int H, int E, int W, int nbits, int my_rank, PeerPtrs dsts,
    LLGeom ll, Parity par, int32_t *__restrict__ num_tokens_per_rank, int32_t *__restrict__ num_tokens_per_expert,
    int32_t *__restrict__ is_token_in_rank, int32_t *__restrict__ send_token_idx_small, int32_t *__restrict__ send_data_offset, int send_waves,
    LLTail tail)
{
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    if (blockIdx.x == 0) {
        layout_small_body<I32, UT>(topk_idx, T, K, E, W, nbits, num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank,
                                   send_token_idx_small, send_data_offset, nullptr, nullptr, smem, 1, 0);
        if (tail.cur_epoch) {
            __syncthreads();                                    // num_tokens_per_expert as this workgroup wrote it
            const uint64_t ep64 = *tail.epoch_bump + 1ull;
            const size_t cpoff = (size_t)(ep64 & 1ull) * tail.counts_parity_stride;
            const int L = tail.L;
            for (int i = threadIdx.x; i < L * W; i += blockDim.x) {
                const int d = i / L, le = i % L;
                sys_store_u64_relaxed((uint64_t *)((uint8_t *)tail.count_peers.p[d] + cpoff) + (size_t)le * W + my_rank,
                                      ((uint64_t)(uint32_t)ep64 << 32) | (uint32_t)num_tokens_per_expert[d * L + le]);
            }
            if (threadIdx.x == 0) *tail.cur_epoch = ep64;
        }
        if (!tail.arrive) return;
    } else {
        const int npairs = T * K;
        for (int i = threadIdx.x; i < npairs; i += blockDim.x) {
            const long long e = ld_idx<I32>(topk_idx, i);
            smem[i] = (e >= 0 && e < E) ? (int32_t)e : -1;
        }
        __syncthreads();
        // kLLSendWaves send waves per workgroup (the rest only helped to load the table): 1024 pairs on 64 workgroups of 16 sending waves left three
        // quarters of the CUs idle and ran 12.4 us; spread over 256 workgroups the rows stream from all of them
        const int wave = (int)(threadIdx.x / kWave);
        const int wid = (int)(blockIdx.x - 1) * send_waves + wave;
        if (wave < send_waves && wid < npairs) {
            const LateIdx late{smem, wid % K, tail.cur_epoch ? ll_row_tag(*tail.epoch_bump + 1ull) : 0u};
            if (QM == MI_EP_QUANT_NONE)
                stage_bf16_body<I32, true>(x, topk_idx, nullptr, nullptr, T, K, H, E, my_rank, dsts, ll, K, (size_t)0, PushGeom{0, 0}, par, wid, late);
            else
                stage_int8_body<I32, QM == MI_EP_QUANT_NONE ? MI_EP_QUANT_INT8 : QM, true>(x, topk_idx, nullptr, nullptr, T, K, H, E, my_rank, dsts, ll, K,
                                                                                            (size_t)0, PushGeom{0, 0}, par, wid, late);
        }
        if (!tail.arrive) return;
    }
    // ---- tail: count this workgroup in; the last one runs the count exchange
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave: its (write-through) stores are performed
    __syncthreads();
    uint32_t *const last_s = (uint32_t *)smem;                  // (the routing table / layout tables are done with)
    if (threadIdx.x == 0) {
        // the layout workgroup's tables were plain stores: release them to the device (num_tokens_per_expert is read by the tail, possibly
        // on another XCD; the other tables by later launches)
        if (blockIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const uint32_t old = __hip_atomic_fetch_add(tail.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_s = old == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (*(volatile uint32_t *)last_s == 0) return;
    __syncthreads();                                            // (everybody has read the word before the body reuses the LDS)
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // num_tokens_per_expert as the layout workgroup wrote it
    __syncthreads();
    const uint64_t ep64 = *tail.epoch_bump + 1ull;
    ll_counts_body(smem, tail.count_peers, num_tokens_per_expert, my_rank, tail.granules_base, ep64, tail.counts_parity_stride, tail.epoch_bump,
                   tail.L, W, tail.count_type, tail.layout_range, tail.packed_recv_count, tail.status, tail.timeout_ticks);
    if (threadIdx.x == 0) __hip_atomic_store(tail.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// pull
// ---------------------------------------------------------------------------------------------
constexpr int kPullWaves = 4;
constexpr int kPullRowsPerBlock = kPullWaves;              // grid sizing: one row per wave until the chip is full (2048 workgroups)

// rows [0, total) of the receive buffers from the (local expert, source) segments described by the inclusive cumsum in LDS
__device__ __forceinline__ void pull_body(
    const PeerPtrs &srcs, const int32_t *cum /*LDS [LW]*/, const int32_t *__restrict__ pull_offset, int seg_capacity, int W, int LW,
    int payload_bytes /*H or 2H*/, uint8_t *__restrict__ recv_x, float *__restrict__ recv_scales, int32_t *__restrict__ recv_src_idx,
    int row_capacity, size_t poff)
{
    const int total = min(cum[LW - 1], row_capacity);      // never write past the caller's buffers
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const size_t stride = MI_EP_ROW_STRIDE(payload_bytes);
    const int n16 = payload_bytes / 16;
    // rows are dealt to waves round-robin over the whole grid: a decode-size exchange (1 K rows) still spreads over every CU,
    // a prefill-size one gives each wave a few rows a grid-width apart
    {
#pragma unroll 1
        for (long long r = (long long)blockIdx.x * kPullWaves + wave; r < total; r += (long long)gridDim.x * kPullWaves) {
            // first i with cum[i] > r
            int lo = 0, hi = LW - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cum[mid] > r) hi = mid; else lo = mid + 1;
            }
            const int i = lo;
            const int j = (int)(r - (i ? cum[i - 1] : 0));
            const int src = i % W;
            const size_t off = pull_offset ? (size_t)pull_offset[i] : (size_t)i * seg_capacity;
            const uint8_t *srow = (const uint8_t *)srcs.p[src] + poff + (off + j) * stride;
            const u32x4 *s16 = (const u32x4 *)srow;
            u32x4 *d16 = (u32x4 *)(recv_x + (size_t)r * payload_bytes);
            copy_row<true, false>(s16, d16, n16, lane);           // straight-line groups of 1 KB pieces (ep_common.h)
            if (lane == 0) {
                const u32x4 m = *(const u32x4 *)(srow + payload_bytes);
                if (recv_scales) recv_scales[r] = __uint_as_float(m[0]);
                recv_src_idx[r * 3 + 0] = (int32_t)m[3];
                recv_src_idx[r * 3 + 1] = (int32_t)m[1];
                recv_src_idx[r * 3 + 2] = (int32_t)m[2];
            }
        }
    }
}


// Store of a received row.  The first kPullWriteBack bytes of a receive buffer are written normally (write-back: the MALL absorbs them),
// everything behind them around the cache (nontemporal).  Left to write-back entirely, the 235 MB of a C2 receive sit dirty in L2 / MALL
// when the pull ends and are evicted under the NEXT kernels' loads (pull 44 us, but the following stage kernel 30 instead of 18 us and
// the combine reduce 110 instead of 102 us); written around entirely, the pull pays for all of its writes itself (52 us).  Share
// written back, C2, one box (tools/probes/nt_from_sweep.sh): 0 MB pull 52.6 / stage 17.7 us, 32 MB 46.8 / 17.6, 64 MB 43.0 / 17.7,
// 96 MB 42.5 / 18.1, 112 MB 42.5 / 19.8, 160 MB 44.8 / 28.9 -- step 0.197 -> 0.182 ms at 64-96 MB.  Smaller buffers (2048 tokens: 117 MB)
// behave the same way: all write-back 24.0 us, all written around 27.5 us.
constexpr size_t kPullWriteBack = 80u << 20;
template <bool NT>
__device__ __forceinline__ void st_row(u32x4 *p, const u32x4 &v)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// first row that is written around the cache (rows are `payload` bytes); INT_MAX = none.  MI_EP_PULL_NT=0 / 1 force none / all,
// MI_EP_PULL_NT_FROM_MB sets the write-back share (measurement only)
static int pull_nt_from_row(long long rows, int payload)
{
    static const char *env = getenv("MI_EP_PULL_NT");
    static const char *env_mb = getenv("MI_EP_PULL_NT_FROM_MB");
    if (env && *env) return atoi(env) != 0 ? 0 : 0x7fffffff;
    const size_t share = env_mb && *env_mb ? (size_t)atoll(env_mb) << 20 : kPullWriteBack;
    const long long from = (long long)(share / (size_t)payload);
    return from >= rows ? 0x7fffffff : (int)from;
}

__global__ __launch_bounds__(kWave * kPullWaves) void pull_kernel(
    PeerPtrs srcs, const int32_t *__restrict__ recv_count, const int32_t *__restrict__ pull_offset, int seg_capacity,
    int W, int LW, int payload_bytes, uint8_t *__restrict__ recv_x, float *__restrict__ recv_scales,
    int32_t *__restrict__ recv_src_idx, int row_capacity, Parity par)
{
    extern __shared__ __attribute__((aligned(16))) int32_t cum[];    // [LW] inclusive cumsum
    for (int i = threadIdx.x; i < LW; i += blockDim.x) cum[i] = recv_count[i];
    __syncthreads();
    pull_body(srcs, cum, pull_offset, seg_capacity, W, LW, payload_bytes, recv_x, recv_scales, recv_src_idx, row_capacity,
              parity_off(par));
}

// The packing launch of the TAGGED low-latency dispatch: every workgroup collects the L*W count granules itself (they were posted at the head
// of the peers' send launches), scans them, and packs rows [0, total) -- waiting (bounded) for the tag in each row's meta word before it
// copies the row.  Workgroup 0 writes the tables and completes the call counter (nothing in this launch reads it: epoch and ping-pong halves
// come from *cur_epoch, left by this rank's own send launch).
__global__ __launch_bounds__(kWave * kPullWaves) void ll_wait_pack_kernel(
    const uint8_t *__restrict__ my_rows, const uint64_t *__restrict__ granules_base, size_t counts_parity_stride, size_t rows_parity_stride,
    const uint64_t *__restrict__ cur_epoch, uint64_t *epoch_bump, int seg_capacity, int W, int L, int payload_bytes, int count_type,
    uint8_t *__restrict__ recv_x, float *__restrict__ recv_scales, int32_t *__restrict__ recv_src_idx, int32_t *__restrict__ layout_range,
    int64_t *__restrict__ packed_recv_count, int row_capacity, int32_t *status, uint64_t timeout_ticks)
{
    extern __shared__ __attribute__((aligned(16))) int32_t c[];       // [LW] counts -> inclusive cumsum, [16] wave totals
    const int LW = L * W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t ep64 = *cur_epoch;
    const uint32_t epoch = (uint32_t)ep64, tag = ll_row_tag(ep64);
    const uint64_t *granules = (const uint64_t *)((const uint8_t *)granules_base + (size_t)(ep64 & 1ull) * counts_parity_stride);
    const uint64_t t0 = ticks_100mhz();
    for (int i = tid; i < LW; i += blockDim.x) {
        uint64_t g;
        while (((g = sys_load_u64(granules + i)) >> 32) != epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                if (blockIdx.x == 0) report_status(status, 2000 + i);
                g = 0;
                break;
            }
        }
        c[i] = (int32_t)(uint32_t)g;
    }
    __syncthreads();
    {   // inclusive scan of c[0..LW) (as ll_counts_body)
        int32_t *wave_tot = c + LW;
        const int per = (LW + (int)blockDim.x - 1) / (int)blockDim.x;
        const int b0 = min(LW, tid * per), b1 = min(LW, b0 + per);
        int32_t sum = 0;
        for (int i = b0; i < b1; ++i) sum += c[i];
        const int32_t inc = wave_incl_scan_i32(sum);
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        int32_t run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wave_tot[w];
        for (int i = b0; i < b1; ++i) {
            run += c[i];
            c[i] = run;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0) {
        for (int i = tid; i < LW; i += blockDim.x) layout_range[i] = c[i];
        for (int le = tid; le < L; le += blockDim.x) {
            const int32_t end = c[(le + 1) * W - 1], beg = le ? c[le * W - 1] : 0;
            packed_recv_count[le] = (count_type == 0) ? (int64_t)end : (int64_t)(end - beg);
        }
        if (tid == 0) *epoch_bump = ep64;
    }
    // rows (pull_body with the tag wait in front of every row)
    const int total = min(c[LW - 1], row_capacity);
    const size_t stride = MI_EP_ROW_STRIDE(payload_bytes), poff = (size_t)(ep64 & 1ull) * rows_parity_stride;
    const int n16 = payload_bytes / 16;
#pragma unroll 1
    for (long long r = (long long)blockIdx.x * kPullWaves + wave; r < total; r += (long long)gridDim.x * kPullWaves) {
        int lo = 0, hi = LW - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (c[mid] > r) hi = mid; else lo = mid + 1;
        }
        const int i = lo;
        const int j = (int)(r - (i ? c[i - 1] : 0));
        const uint8_t *srow = my_rows + poff + ((size_t)i * seg_capacity + j) * stride;
        if (lane == 0) {
            const uint32_t *m3 = (const uint32_t *)(srow + payload_bytes) + 3;
            while ((__hip_atomic_load(m3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> 8) != tag) {
                __builtin_amdgcn_s_sleep(2);
                if (ticks_100mhz() - t0 > timeout_ticks) {
                    report_status(status, 2500 + i % 400);
                    break;
                }
            }
        }
        asm volatile("" ::: "memory");                     // the wave reconverges behind lane 0's wait: the row is read after its tag was seen
        // (system-scope loads, like the poll: a line of this slab half kept by some cache since the call before last must not answer)
        copy_row_sys<false>(srow, (u32x4 *)(recv_x + (size_t)r * payload_bytes), n16, lane);
        if (lane == 0) {
            const uint32_t *mw = (const uint32_t *)(srow + payload_bytes);      // (read like the tag: past every cache)
            u32x4 m;
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = __hip_atomic_load(mw + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (recv_scales) recv_scales[r] = __uint_as_float(m[0]);
            recv_src_idx[r * 3 + 0] = (int32_t)(m[3] & 0xFFu);
            recv_src_idx[r * 3 + 1] = (int32_t)m[1];
            recv_src_idx[r * 3 + 2] = (int32_t)m[2];
        }
    }
}

// pull for compact staging (mi_ep_dispatch_stage_compact): output row r of segment (le, src), position j, is token row
// index[pull_offset + j].t of rank src.  The index entry of a wave's next row is requested before the current row is copied,
// so the extra dependent (possibly remote) read is off the critical path.  Token rows are read up to K times (once per
// selected expert): plain loads, so the local ones come from L2 / MALL after the first touch.
template <bool NT>
__global__ __launch_bounds__(kWave * kPullWaves) void pull_indexed_kernel(
    PeerPtrs srcs, const int32_t *__restrict__ recv_count, const int32_t *__restrict__ pull_offset, int W, int LW,
    int payload_bytes, size_t idx_off, size_t idx_entries, uint8_t *__restrict__ recv_x, float *__restrict__ recv_scales,
    int32_t *__restrict__ recv_src_idx, int row_capacity, Parity par, int skip_src, int nt_from_row)
{
    extern __shared__ __attribute__((aligned(16))) int32_t cum[];    // [LW] inclusive cumsum
    for (int i = threadIdx.x; i < LW; i += blockDim.x) cum[i] = recv_count[i];
    __syncthreads();
    const size_t poff = parity_off(par);
    const int total = min(cum[LW - 1], row_capacity);
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const size_t stride = MI_EP_ROW_STRIDE(payload_bytes);
    const int n16 = payload_bytes / 16;
    const long long nw = (long long)gridDim.x * kPullWaves;
    auto entry = [&](long long r, int &src) -> uint2 {          // wave-uniform
        int lo = 0, hi = LW - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] > r) hi = mid; else lo = mid + 1;
        }
        const int j = (int)(r - (lo ? cum[lo - 1] : 0));
        src = lo % W;
        if (src == skip_src) return uint2{0u, 0u};                // rows of this rank's own tokens: written by pull_local_kernel
        // a corrupt count / offset must not turn into a wild (possibly cross-GPU) read: the index holds idx_entries entries
        const size_t pos = min((size_t)max(pull_offset[lo] + j, 0), idx_entries - 1);
        return ((const uint2 *)((const uint8_t *)srcs.p[src] + poff + idx_off))[pos];
    };
    long long r = (long long)blockIdx.x * kPullWaves + wave;
    int src_n = 0;
    uint2 e_n = uint2{0u, 0u};
    if (r < total) e_n = entry(r, src_n);
#pragma unroll 1
    while (r < total) {
        const int src = src_n;
        const uint2 e = e_n;
        const long long rn = r + nw;
        if (rn < total) e_n = entry(rn, src_n);                  // in flight while this row is copied
        if (src == skip_src) {
            r = rn;
            continue;
        }
        // a stale index entry must not turn into a wild read: token rows live below the index
        const size_t trow = min((size_t)e.x, idx_off / stride - 1);
        const uint8_t *srow = (const uint8_t *)srcs.p[src] + poff + trow * stride;
        const u32x4 *s16 = (const u32x4 *)srow;
        u32x4 *d16 = (u32x4 *)(recv_x + (size_t)r * payload_bytes);
        // straight-line groups of 1 KB pieces (ep_common.h); plain loads: a token row is read up to K times
        if (NT && r >= nt_from_row) copy_row<false, true>(s16, d16, n16, lane);
        else copy_row<false, false>(s16, d16, n16, lane);
        if (lane == 0) {
            if (recv_scales) recv_scales[r] = *(const float *)(srow + payload_bytes);
            recv_src_idx[r * 3 + 0] = src;
            recv_src_idx[r * 3 + 1] = (int32_t)e.x;
            recv_src_idx[r * 3 + 2] = (int32_t)e.y;
        }
        r = rn;
    }
}

// pull_indexed_kernel WITHOUT the copy: where does receive row r live?  One LANE per row: its (source, token row) through the same search and
// index entry, written as the byte offset of the staged row from `base` (the lowest source base; the ping-pong half is part of the offset)
// together with the row's scale and (src, t, k) triple.  The grouped GEMM of fused_deep_moe then reads the staged rows in place
// (mi_ep_moe_gemm1_swiglu_rows): a token's row is staged once and read by its K selections out of L2 / the memory-side cache, and the
// K-fold copy (235 MB written and read back at 4096 tokens x top-8) is never made.  Sources must be LOCAL memory within 4 GiB of `base`.
__global__ __launch_bounds__(256) void resolve_rows_kernel(
    PeerPtrs srcs, const uint8_t *__restrict__ base, const int32_t *__restrict__ recv_count, const int32_t *__restrict__ pull_offset, int W,
    int LW, int payload_bytes, size_t idx_off, size_t idx_entries, uint32_t *__restrict__ row_off, float *__restrict__ recv_scales,
    int32_t *__restrict__ recv_src_idx, int row_capacity, Parity par)
{
    extern __shared__ __attribute__((aligned(16))) int32_t cum[];    // [LW] inclusive cumsum
    for (int i = threadIdx.x; i < LW; i += blockDim.x) cum[i] = recv_count[i];
    __syncthreads();
    const size_t poff = parity_off(par);
    const int total = min(cum[LW - 1], row_capacity);
    const size_t stride = MI_EP_ROW_STRIDE(payload_bytes);
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < row_capacity; r += (long long)gridDim.x * blockDim.x) {
        if (r >= total) {                                        // rows behind the total are never multiplied; keep their table entries tame
            row_off[r] = 0;
            continue;
        }
        int lo = 0, hi = LW - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] > r) hi = mid; else lo = mid + 1;
        }
        const int j = (int)(r - (lo ? cum[lo - 1] : 0));
        const int src = lo % W;
        const uint8_t *sb = (const uint8_t *)srcs.p[src] + poff;
        // a corrupt count / offset / entry must not turn into a wild read: the index holds idx_entries entries, token rows live below it
        const size_t pos = min((size_t)max(pull_offset[lo] + j, 0), idx_entries - 1);
        const uint2 e = ((const uint2 *)(sb + idx_off))[pos];
        const size_t trow = min((size_t)e.x, idx_off / stride - 1);
        const uint8_t *srow = sb + trow * stride;
        row_off[r] = (uint32_t)(size_t)(srow - base);
        if (recv_scales) recv_scales[r] = *(const float *)(srow + payload_bytes);
        recv_src_idx[r * 3 + 0] = src;
        recv_src_idx[r * 3 + 1] = (int32_t)e.x;
        recv_src_idx[r * 3 + 2] = (int32_t)e.y;
    }
}

// Receive rows whose token lives on THIS rank, written token by token instead of row by row: the staged row of token t (own region
// or own source slab) is read ONCE and stored to each of its selections served by this rank's experts -- output row
// start(le, me) + send_token_idx_small[t, k], start = recv_count[le * W + me] - num_tokens_per_expert[me * L + le] -- with the same
// bytes, scale and (me, t, k) triple pull_indexed_kernel produces for it (which then skips source `me`).  At EP = 1 that is the whole
// pull: T staged rows read instead of T * K.
template <bool I32, bool NT>
__global__ __launch_bounds__(kWave * kPullWaves) void pull_local_kernel(
    const uint8_t *__restrict__ my_rows, const void *__restrict__ topk_idx, const int32_t *__restrict__ idx_small,
    const int32_t *__restrict__ recv_count, const int32_t *__restrict__ tokens_per_expert, int T, int K, int E, int W, int my_rank,
    int payload_bytes, uint8_t *__restrict__ recv_x, float *__restrict__ recv_scales, int32_t *__restrict__ recv_src_idx,
    int row_capacity, Parity par, int nt_from_row, int32_t *__restrict__ local_row_out)
{
    const int lane = lane_id();
    const int t = __builtin_amdgcn_readfirstlane(blockIdx.x * kPullWaves + threadIdx.x / kWave);
    if (t >= T) return;
    // the first part of the row is requested before the routing is known (a token without a selection on this rank is the rare case at
    // the sizes where this kernel matters): the routing costs three dependent loads
    const size_t stride = MI_EP_ROW_STRIDE(payload_bytes);
    const uint8_t *srow = my_rows + parity_off(par) + (size_t)t * stride;
    const u32x4 *s16 = (const u32x4 *)srow;
    const int n16 = payload_bytes / 16;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int item = u * kWave + lane;
        if (item < n16) v[u] = s16[item];
    }
    const float scale = *(const float *)(srow + payload_bytes);
    const uint32_t L = (uint32_t)(E / W);
    int r_l = -1;
    if (lane < K) {
        const long long e64 = ld_idx<I32>(topk_idx, (long long)t * K + lane);
        if (e64 >= 0 && e64 < E) {
            const uint32_t e = (uint32_t)e64, rank_e = e / L;     // 32-bit: the 64-bit division is ~150 VALU operations
            if ((int)rank_e == my_rank) {
                const int le = (int)(e - rank_e * L);
                const int r = recv_count[le * W + my_rank] - tokens_per_expert[e] + idx_small[(long long)t * K + lane];
                if (r >= 0 && r < row_capacity) r_l = r;
            }
        }
    }
    unsigned long long lmask = __ballot(r_l >= 0);
    if (lmask == 0ull) return;
    for (int base = 0; base < n16; base += kWave * 8) {
        if (base) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int item = base + u * kWave + lane;
                if (item < n16) v[u] = s16[item];
            }
        }
        for (unsigned long long m = lmask; m; m &= m - 1) {       // wave-uniform walk over this rank's selections of the token
            const int k = __builtin_ctzll(m);
            const int rk = __builtin_amdgcn_readlane(r_l, k);
            u32x4 *d16 = (u32x4 *)(recv_x + (size_t)rk * payload_bytes);
            if (NT && rk >= nt_from_row) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int item = base + u * kWave + lane;
                    if (item < n16) st_row<true>(d16 + item, v[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int item = base + u * kWave + lane;
                    if (item < n16) st_row<false>(d16 + item, v[u]);
                }
            }
        }
    }
    if (r_l >= 0) {                                               // lane k writes the meta of selection k
        if (recv_scales) recv_scales[r_l] = scale;
        recv_src_idx[(size_t)r_l * 3 + 0] = my_rank;
        recv_src_idx[(size_t)r_l * 3 + 1] = t;
        recv_src_idx[(size_t)r_l * 3 + 2] = lane;
        // the receive row of selection (t, k): what mi_ep_combine_push would record for it later (see mi_ep.h)
        if (local_row_out) local_row_out[(size_t)t * K + lane] = r_l;
    }
}

}  // namespace mi_ep

using namespace mi_ep;

extern "C" size_t mi_ep_dispatch_row_bytes(int hidden, int quant_mode)
{
    return MI_EP_ROW_STRIDE((size_t)hidden * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1));
}

extern "C" size_t mi_ep_dispatch_index_offset(int hidden, int quant_mode, int topk, size_t region_bytes)
{
    const size_t rb = mi_ep_dispatch_row_bytes(hidden, quant_mode);
    const size_t cap = region_bytes / (rb + (size_t)topk * 8);      // tokens the region holds with their topk index entries
    return cap * rb;
}

static int stage_launch(const void *x, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                        const int32_t *send_data_offset, int T, int K, int H, int E, int my_rank, int quant_mode, void *rows,
                        size_t idx_off, void *stream, const PeerPtrs *push_peers = nullptr, PushGeom pg = PushGeom{0, 0},
                        Parity par = Parity{EpochRef{nullptr, 0}, 0})
{
    if (T < 0 || K <= 0 || K > MI_EP_MAX_TOPK || H <= 0 || H % 16 || H > MI_EP_MAX_HIDDEN || E <= 0) return MI_EP_EINVAL;
    if (T == 0) return MI_EP_OK;
    if (!x || !topk_idx || !send_token_idx_small || !send_data_offset || (!rows && !push_peers)) return MI_EP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int ksplit = (T <= 512 && !idx_off) ? K : 1;   // decode-size batches: one wave per (token, k) instead of per token
    const int blocks = (int)(((long long)T * ksplit + kStageWaves - 1) / kStageWaves);
    const int threads = kWave * kStageWaves;
    const uint16_t *xp = (const uint16_t *)x;
    uint8_t *rp = (uint8_t *)rows;
    PeerPtrs pp;
    if (push_peers) pp = *push_peers;
    else pp.p[0] = rp;
    const LLGeom ll{0, 0, 0};
#define MI_EP_STAGE(KERNEL) \
    KERNEL<<<blocks, threads, 0, s>>>(xp, topk_idx, send_token_idx_small, send_data_offset, T, K, H, E, my_rank, pp, ll, ksplit, idx_off, pg, par)
    switch (quant_mode) {
        case MI_EP_QUANT_NONE:
            if (idx_is_i32) MI_EP_STAGE(stage_bf16_kernel<true>); else MI_EP_STAGE(stage_bf16_kernel<false>);
            break;
        case MI_EP_QUANT_INT8:
            if (idx_is_i32) MI_EP_STAGE((stage_int8_kernel<true, MI_EP_QUANT_INT8>)); else MI_EP_STAGE((stage_int8_kernel<false, MI_EP_QUANT_INT8>));
            break;
        case MI_EP_QUANT_INT8_NOEPS:
            if (idx_is_i32) MI_EP_STAGE((stage_int8_kernel<true, MI_EP_QUANT_INT8_NOEPS>)); else MI_EP_STAGE((stage_int8_kernel<false, MI_EP_QUANT_INT8_NOEPS>));
            break;
        case MI_EP_QUANT_FP8_E4M3:
            if (idx_is_i32) MI_EP_STAGE((stage_int8_kernel<true, MI_EP_QUANT_FP8_E4M3>)); else MI_EP_STAGE((stage_int8_kernel<false, MI_EP_QUANT_FP8_E4M3>));
            break;
        default:
            return MI_EP_EINVAL;
    }
#undef MI_EP_STAGE
    return launch_status();
}

extern "C" int mi_ep_dispatch_stage(const void *x, const void *topk_idx, int idx_is_i32,
                                    const int32_t *send_token_idx_small, const int32_t *send_data_offset, int T, int K,
                                    int H, int E, int my_rank, int quant_mode, void *rows, void *stream)
{
    return stage_launch(x, topk_idx, idx_is_i32, send_token_idx_small, send_data_offset, T, K, H, E, my_rank, quant_mode, rows, 0,
                        stream);
}

extern "C" int mi_ep_dispatch_stage_compact(const void *x, const void *topk_idx, int idx_is_i32,
                                            const int32_t *send_token_idx_small, const int32_t *send_data_offset, int T, int K,
                                            int H, int E, int my_rank, int quant_mode, void *region, size_t region_bytes,
                                            const uint64_t *epoch_ctr, size_t parity_stride, void *stream)
{
    if (H <= 0 || K <= 0) return MI_EP_EINVAL;
    const size_t rb = mi_ep_dispatch_row_bytes(H, quant_mode);
    const size_t idx_off = mi_ep_dispatch_index_offset(H, quant_mode, K, region_bytes);
    if (idx_off == 0 || (size_t)T > idx_off / rb) return MI_EP_EINVAL;          // region too small for T tokens
    return stage_launch(x, topk_idx, idx_is_i32, send_token_idx_small, send_data_offset, T, K, H, E, my_rank, quant_mode, region,
                        idx_off, stream, nullptr, PushGeom{0, 0}, make_parity(epoch_ctr, 1, parity_stride));
}

extern "C" size_t mi_ep_dispatch_push_slab_bytes(size_t region_bytes, int num_ranks)
{
    if (num_ranks <= 0) return 0;
    return (region_bytes / (size_t)num_ranks) & ~(size_t)255;
}

extern "C" int mi_ep_dispatch_stage_push(const void *x, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                                         const int32_t *send_data_offset, int T, int K, int H, int E, int W, int my_rank,
                                         int quant_mode, void *const *peer_region_host, size_t region_bytes,
                                         const uint64_t *epoch_ctr, size_t parity_stride, void *stream)
{
    if (H <= 0 || K <= 0 || W <= 0 || W > MI_EP_MAX_RANKS || E <= 0 || E % W || my_rank < 0 || my_rank >= W || !peer_region_host)
        return MI_EP_EINVAL;
    const size_t rb = mi_ep_dispatch_row_bytes(H, quant_mode);
    const size_t slab = mi_ep_dispatch_push_slab_bytes(region_bytes, W);
    const size_t idx_off = mi_ep_dispatch_index_offset(H, quant_mode, K, slab);
    if (idx_off == 0 || (size_t)T > idx_off / rb) return MI_EP_EINVAL;          // a slab cannot hold T tokens
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!peer_region_host[i]) return MI_EP_EINVAL;
        pp.p[i] = peer_region_host[i];
    }
    return stage_launch(x, topk_idx, idx_is_i32, send_token_idx_small, send_data_offset, T, K, H, E, my_rank, quant_mode, nullptr,
                        idx_off, stream, &pp, PushGeom{E / W, slab}, make_parity(epoch_ctr, 1, parity_stride));
}

extern "C" int mi_ep_dispatch_pull(const void *const *src_base_host, const int32_t *recv_count,
                                   const int32_t *pull_offset, int W, int L, int H, int quant_mode, int rows_hint,
                                   void *recv_x, float *recv_x_scales, int32_t *recv_src_idx, void *stream)
{
    if (!src_base_host || !recv_count || !pull_offset || W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || H <= 0 || H % 16 ||
        !recv_x || !recv_src_idx)
        return MI_EP_EINVAL;
    if (rows_hint <= 0) return MI_EP_OK;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!src_base_host[i]) return MI_EP_EINVAL;
        pp.p[i] = const_cast<void *>(src_base_host[i]);
    }
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    long long blocks = ((long long)rows_hint + kPullRowsPerBlock - 1) / kPullRowsPerBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = (size_t)L * W * sizeof(int32_t);
    pull_kernel<<<(int)blocks, kWave * kPullWaves, lds, (hipStream_t)stream>>>(
        pp, recv_count, pull_offset, 0, W, L * W, payload, (uint8_t *)recv_x, recv_x_scales, recv_src_idx, rows_hint,
        Parity{EpochRef{nullptr, 0}, 0});
    return launch_status();
}

extern "C" int mi_ep_dispatch_pull_indexed(const void *const *src_base_host, const int32_t *recv_count,
                                           const int32_t *pull_offset, int W, int L, int H, int K, int quant_mode,
                                           int rows_hint, size_t region_bytes, void *recv_x, float *recv_x_scales,
                                           int32_t *recv_src_idx, const uint64_t *epoch_ctr, size_t parity_stride, int skip_src,
                                           void *stream)
{
    if (!src_base_host || !recv_count || !pull_offset || W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || H <= 0 || H % 16 || K <= 0 ||
        K > MI_EP_MAX_TOPK || !recv_x || !recv_src_idx || skip_src >= W)
        return MI_EP_EINVAL;
    if (rows_hint <= 0) return MI_EP_OK;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!src_base_host[i]) return MI_EP_EINVAL;
        pp.p[i] = const_cast<void *>(src_base_host[i]);
    }
    const size_t idx_off = mi_ep_dispatch_index_offset(H, quant_mode, K, region_bytes);
    if (idx_off == 0) return MI_EP_EINVAL;
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    long long blocks = ((long long)rows_hint + kPullRowsPerBlock - 1) / kPullRowsPerBlock;
    static const long long cap = getenv("MI_EP_PULL_BLOCKS") ? atoll(getenv("MI_EP_PULL_BLOCKS")) : 256 * 8;
    if (blocks > cap) blocks = cap;
    const size_t lds = (size_t)L * W * sizeof(int32_t);
#define MI_EP_PULL_INDEXED(NT)                                                                                                      \
    pull_indexed_kernel<NT><<<(int)blocks, kWave * kPullWaves, lds, (hipStream_t)stream>>>(                                         \
        pp, recv_count, pull_offset, W, L * W, payload, idx_off, (idx_off / mi_ep_dispatch_row_bytes(H, quant_mode)) * (size_t)K,  \
        (uint8_t *)recv_x, recv_x_scales, recv_src_idx, rows_hint, make_parity(epoch_ctr, 0, parity_stride), skip_src < 0 ? -1 : skip_src, \
        nt_from)
    const int nt_from = pull_nt_from_row(rows_hint, payload);
    if (nt_from != 0x7fffffff) MI_EP_PULL_INDEXED(true); else MI_EP_PULL_INDEXED(false);
#undef MI_EP_PULL_INDEXED
    return launch_status();
}

extern "C" int mi_ep_dispatch_resolve_rows(const void *const *src_base_host, const int32_t *recv_count, const int32_t *pull_offset, int W,
                                           int L, int H, int K, int quant_mode, int rows_cap, size_t region_bytes, const void **a_base_out,
                                           uint32_t *row_offsets, float *recv_x_scales, int32_t *recv_src_idx, const uint64_t *epoch_ctr,
                                           size_t parity_stride, void *stream)
{
    if (!src_base_host || !recv_count || !pull_offset || W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || H <= 0 || H % 16 || K <= 0 ||
        K > MI_EP_MAX_TOPK || !row_offsets || !recv_src_idx || !a_base_out)
        return MI_EP_EINVAL;
    const size_t idx_off = mi_ep_dispatch_index_offset(H, quant_mode, K, region_bytes);
    if (idx_off == 0) return MI_EP_EINVAL;
    PeerPtrs pp;
    const uint8_t *lo = nullptr, *hi = nullptr;
    for (int i = 0; i < W; ++i) {
        if (!src_base_host[i]) return MI_EP_EINVAL;
        pp.p[i] = const_cast<void *>(src_base_host[i]);
        const uint8_t *b = (const uint8_t *)src_base_host[i];
        if (!lo || b < lo) lo = b;
        if (!hi || b > hi) hi = b;
    }
    // every staged row (either ping-pong half) must lie within 32 bits of the lowest base
    if ((size_t)(hi - lo) + (epoch_ctr ? parity_stride : 0) + idx_off > 0xFFFFFFFFull) return MI_EP_ESIZE;
    *a_base_out = lo;
    if (rows_cap <= 0) return MI_EP_OK;
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    const int blocks = std::min((rows_cap + 255) / 256, 2048);
    resolve_rows_kernel<<<blocks, 256, (size_t)L * W * sizeof(int32_t), (hipStream_t)stream>>>(
        pp, lo, recv_count, pull_offset, W, L * W, payload, idx_off, (idx_off / mi_ep_dispatch_row_bytes(H, quant_mode)) * (size_t)K, row_offsets,
        recv_x_scales, recv_src_idx, rows_cap, make_parity(epoch_ctr, 0, parity_stride));
    return launch_status();
}

extern "C" int mi_ep_dispatch_pull_local(const void *my_rows, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                                         const int32_t *recv_count, const int32_t *num_tokens_per_expert, int T, int K, int H, int E,
                                         int W, int my_rank, int quant_mode, int rows_hint, void *recv_x, float *recv_x_scales,
                                         int32_t *recv_src_idx, int32_t *local_row_out, const uint64_t *epoch_ctr, size_t parity_stride,
                                         void *stream)
{
    if (T < 0 || K <= 0 || K > MI_EP_MAX_TOPK || H <= 0 || H % 16 || E <= 0 || W <= 0 || W > MI_EP_MAX_RANKS || E % W || my_rank < 0 ||
        my_rank >= W)
        return MI_EP_EINVAL;
    if (T == 0 || rows_hint <= 0) return MI_EP_OK;               // a rank without tokens has nothing of its own to gather
    if (!my_rows || !topk_idx || !send_token_idx_small || !recv_count || !num_tokens_per_expert || !recv_x || !recv_src_idx)
        return MI_EP_EINVAL;
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    const int blocks = (T + kPullWaves - 1) / kPullWaves;
    const Parity par = make_parity(epoch_ctr, 0, parity_stride);
    const int nt_from = pull_nt_from_row(rows_hint, payload);
    const bool nt = nt_from != 0x7fffffff;
#define MI_EP_PULL_LOCAL(I32)                                                                                                       \
    if (nt) MI_EP_PULL_LOCAL2(I32, true); else MI_EP_PULL_LOCAL2(I32, false)
#define MI_EP_PULL_LOCAL2(I32, NT)                                                                                                  \
    pull_local_kernel<I32, NT><<<blocks, kWave * kPullWaves, 0, (hipStream_t)stream>>>(                                                \
        (const uint8_t *)my_rows, topk_idx, send_token_idx_small, recv_count, num_tokens_per_expert, T, K, E, W, my_rank, payload, \
        (uint8_t *)recv_x, recv_x_scales, recv_src_idx, rows_hint, par, nt_from, local_row_out)
    if (idx_is_i32) { MI_EP_PULL_LOCAL(true); } else { MI_EP_PULL_LOCAL(false); }
#undef MI_EP_PULL_LOCAL
#undef MI_EP_PULL_LOCAL2
    return launch_status();
}

// ---------------------------------------------------------------------------------------------
// A5 low-latency dispatch (same stage / pull kernels, different geometry; no host sync anywhere)
// ---------------------------------------------------------------------------------------------
namespace mi_ep {

__global__ void ll_post_counts_kernel(PeerPtrs peers, const int32_t *__restrict__ cnt, int L, int W, int my_rank,
                                      uint32_t epoch)
{
    const int d = blockIdx.x;
    uint64_t *g = (uint64_t *)peers.p[d];
    for (int le = threadIdx.x; le < L; le += blockDim.x)
        sys_store_u64_relaxed(g + (size_t)le * W + my_rank, ((uint64_t)epoch << 32) | (uint32_t)cnt[d * L + le]);
}

// one workgroup: wait for the L*W count granules, inclusive cumsum in idx-i order, per-expert counts (ll_counts_body, above)
__global__ __launch_bounds__(256) void ll_counts_kernel(PeerPtrs count_peers, const int32_t *__restrict__ my_counts_out /*[E] or null*/,
                                                        int my_rank, const uint64_t *__restrict__ granules_base, EpochRef er,
                                                        size_t counts_parity_stride, uint64_t *epoch_bump, int L, int W,
                                                        int count_type, int32_t *__restrict__ layout_range,
                                                        int64_t *__restrict__ packed_recv_count, int32_t *status,
                                                        uint64_t timeout_ticks)
{
    extern __shared__ __attribute__((aligned(16))) int32_t c[];   // [L*W] counts -> inclusive cumsum, then [16] wave totals
    ll_counts_body(c, count_peers, my_counts_out, my_rank, granules_base, epoch_of(er), counts_parity_stride, epoch_bump, L, W, count_type,
                   layout_range, packed_recv_count, status, timeout_ticks);
}

}  // namespace mi_ep

extern "C" int mi_ep_ll_dispatch_send(const void *x, const void *topk_idx, int idx_is_i32,
                                      const int32_t *send_token_idx_small, int T, int K, int H, int E, int W, int my_rank,
                                      int max_tokens, int quant_mode, void *const *peer_rows_host, const uint64_t *epoch_ctr,
                                      size_t parity_stride, void *stream)
{
    if (T < 0 || K <= 0 || K > MI_EP_MAX_TOPK || H <= 0 || H % 16 || H > MI_EP_MAX_HIDDEN || E <= 0 || W <= 0 ||
        W > MI_EP_MAX_RANKS || E % W || T > max_tokens || !peer_rows_host)
        return MI_EP_EINVAL;
    if (T == 0) return MI_EP_OK;
    if (!x || !topk_idx || !send_token_idx_small) return MI_EP_EINVAL;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!peer_rows_host[i]) return MI_EP_EINVAL;
        pp.p[i] = peer_rows_host[i];
    }
    const LLGeom ll{E / W, W, max_tokens};
    const Parity par = make_parity(epoch_ctr, 1, parity_stride);
    hipStream_t s = (hipStream_t)stream;
    const int ksplit = T <= 512 ? K : 1;            // decode-size batches: one wave per (token, k) instead of per token
    const int blocks = (int)(((long long)T * ksplit + kStageWaves - 1) / kStageWaves);
    const int threads = kWave * kStageWaves;
    const uint16_t *xp = (const uint16_t *)x;
#define MI_EP_STAGE(KERNEL) \
    KERNEL<<<blocks, threads, 0, s>>>(xp, topk_idx, send_token_idx_small, nullptr, T, K, H, E, my_rank, pp, ll, ksplit, (size_t)0, PushGeom{0, 0}, par)
    switch (quant_mode) {
        case MI_EP_QUANT_NONE:
            if (idx_is_i32) MI_EP_STAGE(stage_bf16_kernel<true>); else MI_EP_STAGE(stage_bf16_kernel<false>);
            break;
        case MI_EP_QUANT_INT8:
            if (idx_is_i32) MI_EP_STAGE((stage_int8_kernel<true, MI_EP_QUANT_INT8>)); else MI_EP_STAGE((stage_int8_kernel<false, MI_EP_QUANT_INT8>));
            break;
        case MI_EP_QUANT_INT8_NOEPS:
            if (idx_is_i32) MI_EP_STAGE((stage_int8_kernel<true, MI_EP_QUANT_INT8_NOEPS>)); else MI_EP_STAGE((stage_int8_kernel<false, MI_EP_QUANT_INT8_NOEPS>));
            break;
        case MI_EP_QUANT_FP8_E4M3:
            if (idx_is_i32) MI_EP_STAGE((stage_int8_kernel<true, MI_EP_QUANT_FP8_E4M3>)); else MI_EP_STAGE((stage_int8_kernel<false, MI_EP_QUANT_FP8_E4M3>));
            break;
        default:
            return MI_EP_EINVAL;
    }
#undef MI_EP_STAGE
    return launch_status();
}

static int ll_layout_send_launch(const void *x, const void *topk_idx, int idx_is_i32, int T, int K, int H, int E, int W, int my_rank,
                                 int max_tokens, int quant_mode, void *const *peer_rows_host, const uint64_t *epoch_ctr,
                                 size_t parity_stride, int32_t *num_tokens_per_rank, int32_t *num_tokens_per_expert,
                                 int32_t *is_token_in_rank, int32_t *send_token_idx_small, int32_t *send_data_offset, const LLTail *tail_in,
                                 void *stream)
{
    if (T < 0 || K <= 0 || K > MI_EP_MAX_TOPK || H <= 0 || H % 16 || H > MI_EP_MAX_HIDDEN || E <= 0 || W <= 0 ||
        W > MI_EP_MAX_RANKS || E % W || T > max_tokens || !peer_rows_host || !num_tokens_per_rank || !num_tokens_per_expert ||
        !send_data_offset)
        return MI_EP_EINVAL;
    // one layout workgroup: at most 16 units of 16 tokens (<= 256 tokens) or of 64 tokens (<= 1024), and the histograms must fit its LDS
    const int ut = T <= 256 ? 16 : kLayoutUnitTokens;
    if ((T + ut - 1) / ut > 16 || (size_t)16 * E > 16384 || ((16 * E) & 1)) return MI_EP_EINVAL;
    if (T > 0 && (!x || !topk_idx || !is_token_in_rank || !send_token_idx_small)) return MI_EP_EINVAL;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!peer_rows_host[i]) return MI_EP_EINVAL;
        pp.p[i] = peer_rows_host[i];
    }
    const LLGeom ll{E / W, W, max_tokens};
    const Parity par = make_parity(epoch_ctr, 1, parity_stride);
    hipStream_t s = (hipStream_t)stream;
    int nbits = 1;
    while ((1 << nbits) < E) ++nbits;
    static const int send_waves_env = getenv("MI_EP_LL_SEND_WAVES") ? atoi(getenv("MI_EP_LL_SEND_WAVES")) : 0;
    const int send_waves = send_waves_env >= 1 && send_waves_env <= 16 ? send_waves_env : kLLSendWaves;
    const int blocks = 1 + (int)(((long long)T * K + send_waves - 1) / send_waves);       // the layout workgroup + the send workgroups
    // dynamic LDS: the layout workgroup's tables, or the send workgroups' copy of the routing table (int32 per pair)
    const LLTail tail = tail_in ? *tail_in : LLTail{};
    // dynamic LDS: the layout workgroup's tables, the send workgroups' copy of the routing table (int32 per pair), the tail's counts
    const size_t lds = std::max(std::max(layout_small_lds_bytes(E, W, ut), (size_t)T * K * sizeof(int32_t)), tail_in ? (size_t)(E + 16) * 4 : (size_t)0);
    const uint16_t *xp = (const uint16_t *)x;
    static PerDeviceOnce attr_once;
#define MI_EP_LLS_ATTR(I32, QM, UT) (void)hipFuncSetAttribute((const void *)ll_layout_send_kernel<I32, QM, UT>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)
    if (attr_once.need()) {
        MI_EP_LLS_ATTR(true, MI_EP_QUANT_NONE, 16); MI_EP_LLS_ATTR(false, MI_EP_QUANT_NONE, 16); MI_EP_LLS_ATTR(true, MI_EP_QUANT_NONE, 64); MI_EP_LLS_ATTR(false, MI_EP_QUANT_NONE, 64);
        MI_EP_LLS_ATTR(true, MI_EP_QUANT_INT8, 16); MI_EP_LLS_ATTR(false, MI_EP_QUANT_INT8, 16); MI_EP_LLS_ATTR(true, MI_EP_QUANT_INT8, 64); MI_EP_LLS_ATTR(false, MI_EP_QUANT_INT8, 64);
        MI_EP_LLS_ATTR(true, MI_EP_QUANT_INT8_NOEPS, 16); MI_EP_LLS_ATTR(false, MI_EP_QUANT_INT8_NOEPS, 16); MI_EP_LLS_ATTR(true, MI_EP_QUANT_INT8_NOEPS, 64); MI_EP_LLS_ATTR(false, MI_EP_QUANT_INT8_NOEPS, 64);
        MI_EP_LLS_ATTR(true, MI_EP_QUANT_FP8_E4M3, 16); MI_EP_LLS_ATTR(false, MI_EP_QUANT_FP8_E4M3, 16); MI_EP_LLS_ATTR(true, MI_EP_QUANT_FP8_E4M3, 64); MI_EP_LLS_ATTR(false, MI_EP_QUANT_FP8_E4M3, 64);
    }
#undef MI_EP_LLS_ATTR
#define MI_EP_LLS(I32, QM, UT)                                                                                                         \
    ll_layout_send_kernel<I32, QM, UT><<<blocks, 1024, lds, s>>>(xp, topk_idx, T, K, H, E, W, nbits, my_rank, pp, ll, par,              \
                                                                  num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank,         \
                                                                  send_token_idx_small, send_data_offset, send_waves, tail)
#define MI_EP_LLS_Q(QM)                                                                                       \
    do {                                                                                                      \
        if (ut == 16) { if (idx_is_i32) MI_EP_LLS(true, QM, 16); else MI_EP_LLS(false, QM, 16); }             \
        else { if (idx_is_i32) MI_EP_LLS(true, QM, 64); else MI_EP_LLS(false, QM, 64); }                      \
    } while (0)
    switch (quant_mode) {
        case MI_EP_QUANT_NONE: MI_EP_LLS_Q(MI_EP_QUANT_NONE); break;
        case MI_EP_QUANT_INT8: MI_EP_LLS_Q(MI_EP_QUANT_INT8); break;
        case MI_EP_QUANT_INT8_NOEPS: MI_EP_LLS_Q(MI_EP_QUANT_INT8_NOEPS); break;
        case MI_EP_QUANT_FP8_E4M3: MI_EP_LLS_Q(MI_EP_QUANT_FP8_E4M3); break;
        default: return MI_EP_EINVAL;
    }
#undef MI_EP_LLS_Q
#undef MI_EP_LLS
    return launch_status();
}

extern "C" int mi_ep_ll_dispatch_layout_send(const void *x, const void *topk_idx, int idx_is_i32, int T, int K, int H, int E, int W, int my_rank,
                                            int max_tokens, int quant_mode, void *const *peer_rows_host, const uint64_t *epoch_ctr,
                                            size_t parity_stride, int32_t *num_tokens_per_rank, int32_t *num_tokens_per_expert,
                                            int32_t *is_token_in_rank, int32_t *send_token_idx_small, int32_t *send_data_offset,
                                            void *stream)
{
    return ll_layout_send_launch(x, topk_idx, idx_is_i32, T, K, H, E, W, my_rank, max_tokens, quant_mode, peer_rows_host, epoch_ctr, parity_stride,
                                 num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank, send_token_idx_small, send_data_offset, nullptr,
                                 stream);
}

// layout + send + the count exchange in ONE launch (the tail of ll_layout_send_kernel); mi_ep_ll_pack then fills the outputs
extern "C" int mi_ep_ll_dispatch_layout_send_counts(const void *x, const void *topk_idx, int idx_is_i32, int T, int K, int H, int E, int W,
                                                   int my_rank, int max_tokens, int quant_mode, void *const *peer_rows_host,
                                                   uint64_t *epoch_ctr, size_t parity_stride, int32_t *num_tokens_per_rank,
                                                   int32_t *num_tokens_per_expert, int32_t *is_token_in_rank, int32_t *send_token_idx_small,
                                                   int32_t *send_data_offset, uint64_t *const *peer_counts_host, const uint64_t *my_counts,
                                                   size_t counts_parity_stride, int count_type, int32_t *layout_range,
                                                   int64_t *packed_recv_count, uint32_t *arrive_word, int32_t *status, int timeout_ms,
                                                   void *stream)
{
    if (!epoch_ctr || !peer_counts_host || !my_counts || !layout_range || !packed_recv_count || !arrive_word || !status || W <= 0 ||
        W > MI_EP_MAX_RANKS || E <= 0 || E % W || E > 2048)
        return MI_EP_EINVAL;
    LLTail tail{};
    tail.arrive = arrive_word, tail.granules_base = my_counts, tail.counts_parity_stride = counts_parity_stride, tail.epoch_bump = epoch_ctr;
    tail.L = E / W, tail.count_type = count_type, tail.layout_range = layout_range, tail.packed_recv_count = packed_recv_count;
    tail.status = status, tail.timeout_ticks = (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull;
    for (int i = 0; i < W; ++i) {
        if (!peer_counts_host[i]) return MI_EP_EINVAL;
        tail.count_peers.p[i] = peer_counts_host[i];
    }
    return ll_layout_send_launch(x, topk_idx, idx_is_i32, T, K, H, E, W, my_rank, max_tokens, quant_mode, peer_rows_host, epoch_ctr, parity_stride,
                                 num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank, send_token_idx_small, send_data_offset, &tail,
                                 stream);
}

// layout + send with TAGGED rows and the counts posted from the head of the launch (no count exchange launch: mi_ep_ll_wait_pack follows)
extern "C" int mi_ep_ll_dispatch_layout_send_tagged(const void *x, const void *topk_idx, int idx_is_i32, int T, int K, int H, int E, int W,
                                                   int my_rank, int max_tokens, int quant_mode, void *const *peer_rows_host,
                                                   const uint64_t *epoch_ctr, size_t parity_stride, int32_t *num_tokens_per_rank,
                                                   int32_t *num_tokens_per_expert, int32_t *is_token_in_rank, int32_t *send_token_idx_small,
                                                   int32_t *send_data_offset, uint64_t *const *peer_counts_host, size_t counts_parity_stride,
                                                   uint64_t *cur_epoch_word, void *stream)
{
    if (!epoch_ctr || !peer_counts_host || !cur_epoch_word || W <= 0 || W > MI_EP_MAX_RANKS || E <= 0 || E % W || E > 2048) return MI_EP_EINVAL;
    LLTail tail{};
    tail.counts_parity_stride = counts_parity_stride, tail.epoch_bump = const_cast<uint64_t *>(epoch_ctr), tail.L = E / W, tail.cur_epoch = cur_epoch_word;
    for (int i = 0; i < W; ++i) {
        if (!peer_counts_host[i]) return MI_EP_EINVAL;
        tail.count_peers.p[i] = peer_counts_host[i];
    }
    return ll_layout_send_launch(x, topk_idx, idx_is_i32, T, K, H, E, W, my_rank, max_tokens, quant_mode, peer_rows_host, epoch_ctr, parity_stride,
                                 num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank, send_token_idx_small, send_data_offset, &tail,
                                 stream);
}

// the packing launch behind mi_ep_ll_dispatch_layout_send_tagged: collects the counts, waits per row for its tag, completes *epoch_ctr
extern "C" int mi_ep_ll_wait_pack(const void *my_rows, const uint64_t *my_counts, size_t counts_parity_stride, int W, int L, int max_tokens, int H,
                                  int quant_mode, int count_type, void *packed_recv_x, float *packed_recv_x_scales, int64_t *packed_recv_count,
                                  int32_t *src_info, int32_t *layout_range, int rows_capacity, const uint64_t *cur_epoch_word, uint64_t *epoch_ctr,
                                  size_t rows_parity_stride, int32_t *status, int timeout_ms, int max_blocks, void *stream)
{
    if (!my_rows || !my_counts || !packed_recv_x || !packed_recv_count || !src_info || !layout_range || !cur_epoch_word || !epoch_ctr || !status ||
        W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || L * W > 2048 || H <= 0 || H % 16 || max_tokens <= 0)
        return MI_EP_EINVAL;
    const int cap = rows_capacity > 0 ? rows_capacity : L * W * max_tokens;
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    // the waves of this launch WAIT for rows: at most 512 workgroups (a quarter of the chip's slots), the rows are grid-strided
    long long blocks = ((long long)L * W * max_tokens + kPullRowsPerBlock - 1) / kPullRowsPerBlock;
    // MI_EP_LL_WAIT_BLOCKS (measurement / shared-GPU setups): with two PROCESSES on one GPU, 512 waiting workgroups of one (two on every CU) keep
    // the other's 1024-thread send workgroups from starting -- (2 ranks, 300 tokens, H = 2048) timed out in half of the runs and took 9-30 s in
    // the others; capped at 64 the same case runs in 2 s every time, and a longer sleep between polls changes nothing (it is placement, not
    // polling traffic).  Ranks that share a GPU therefore default to the three-launch form (deep_ep.hpp); one rank per GPU has no such peer.
    static const long long cap_env = getenv("MI_EP_LL_WAIT_BLOCKS") ? atoll(getenv("MI_EP_LL_WAIT_BLOCKS")) : 512;
    if (blocks > cap_env) blocks = cap_env;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;      // (the caller knows that ranks share this GPU)
    if (blocks < 1) blocks = 1;
    ll_wait_pack_kernel<<<(int)blocks, kWave * kPullWaves, (size_t)(L * W + 16) * 4, (hipStream_t)stream>>>(
        (const uint8_t *)my_rows, my_counts, counts_parity_stride, rows_parity_stride, cur_epoch_word, epoch_ctr, max_tokens, W, L, payload, count_type,
        (uint8_t *)packed_recv_x, packed_recv_x_scales, src_info, layout_range, packed_recv_count, cap, status,
        (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull);
    return launch_status();
}

// the packing half of mi_ep_ll_post_recv on its own: rows from this rank's slabs into the packed outputs, by the cumulative counts the
// count exchange left in layout_range (device); epoch_ctr = the family's call counter, already completed for this call (add = 0)
extern "C" int mi_ep_ll_pack(const void *my_rows, const int32_t *layout_range, int W, int L, int max_tokens, int H, int quant_mode,
                             void *packed_recv_x, float *packed_recv_x_scales, int32_t *src_info, int rows_capacity, const uint64_t *epoch_ctr,
                             size_t rows_parity_stride, void *stream)
{
    if (!my_rows || !layout_range || !packed_recv_x || !src_info || W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || L * W > 2048 || H <= 0 || H % 16 ||
        max_tokens <= 0)
        return MI_EP_EINVAL;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) pp.p[i] = const_cast<void *>(my_rows);
    const int cap = rows_capacity > 0 ? rows_capacity : L * W * max_tokens;      // rows the caller's output buffers hold
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    long long blocks = ((long long)L * W * max_tokens + kPullRowsPerBlock - 1) / kPullRowsPerBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    pull_kernel<<<(int)blocks, kWave * kPullWaves, (size_t)L * W * 4, (hipStream_t)stream>>>(pp, layout_range, nullptr, max_tokens, W, L * W, payload,
                                                                                         (uint8_t *)packed_recv_x, packed_recv_x_scales, src_info,
                                                                                         cap, make_parity(epoch_ctr, 0, rows_parity_stride));
    return launch_status();
}

extern "C" int mi_ep_ll_post_counts(uint64_t *const *peer_counts_host, const int32_t *num_tokens_per_expert, int E, int W,
                                    int my_rank, uint32_t epoch, void *stream)
{
    if (!peer_counts_host || !num_tokens_per_expert || E <= 0 || W <= 0 || W > MI_EP_MAX_RANKS || E % W || epoch == 0)
        return MI_EP_EINVAL;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!peer_counts_host[i]) return MI_EP_EINVAL;
        pp.p[i] = peer_counts_host[i];
    }
    ll_post_counts_kernel<<<W, 64, 0, (hipStream_t)stream>>>(pp, num_tokens_per_expert, E / W, W, my_rank, epoch);
    return launch_status();
}

extern "C" int mi_ep_ll_dispatch_recv(const void *my_rows, const uint64_t *my_counts, uint32_t epoch, int W, int L,
                                      int max_tokens, int H, int quant_mode, int count_type, void *packed_recv_x,
                                      float *packed_recv_x_scales, int64_t *packed_recv_count, int32_t *src_info,
                                      int32_t *layout_range, int rows_capacity, int32_t *status, int timeout_ms, void *stream)
{
    if (!my_rows || !my_counts || !packed_recv_x || !packed_recv_count || !src_info || !layout_range || !status ||
        W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || L * W > 2048 || H <= 0 || H % 16 || max_tokens <= 0 || epoch == 0)
        return MI_EP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const uint64_t ticks = (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull;
    PeerPtrs none{};
    ll_counts_kernel<<<1, 256, (size_t)(L * W + 16) * 4, s>>>(none, nullptr, 0, my_counts, EpochRef{nullptr, epoch}, 0, nullptr, L, W,
                                                            count_type, layout_range, packed_recv_count, status, ticks);
    const int cap = rows_capacity > 0 ? rows_capacity : L * W * max_tokens;      // rows the caller's output buffers hold
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) pp.p[i] = const_cast<void *>(my_rows);
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    // worst case rows = W * max_tokens * min(K, L) is not known here (K); size the grid for W * max_tokens rows per
    // 16-row block, capped: the kernel grid-strides over the device-side total.
    // worst case rows = every (local expert, source) slab full; the kernel grid-strides over the device-side total, so
    // size the grid to fill the chip even for decode-size batches (a 128-token call has ~1 K rows = 7 MB to move)
    long long blocks = ((long long)L * W * max_tokens + kPullRowsPerBlock - 1) / kPullRowsPerBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    pull_kernel<<<(int)blocks, kWave * kPullWaves, (size_t)L * W * 4, s>>>(pp, layout_range, nullptr, max_tokens, W, L * W,
                                                                         payload, (uint8_t *)packed_recv_x,
                                                                         packed_recv_x_scales, src_info, cap,
                                                                         Parity{EpochRef{nullptr, 0}, 0});
    return launch_status();
}

extern "C" int mi_ep_ll_post_recv(uint64_t *const *peer_counts_host, const int32_t *num_tokens_per_expert, int my_rank,
                                  const void *my_rows, const uint64_t *my_counts, uint32_t epoch, int W, int L, int max_tokens, int H,
                                  int quant_mode, int count_type, void *packed_recv_x, float *packed_recv_x_scales,
                                  int64_t *packed_recv_count, int32_t *src_info, int32_t *layout_range, int rows_capacity,
                                  uint64_t *epoch_ctr, size_t rows_parity_stride, size_t counts_parity_stride, int32_t *status,
                                  int timeout_ms, void *stream)
{
    if (!peer_counts_host || !num_tokens_per_expert || !my_rows || !my_counts || !packed_recv_x || !packed_recv_count || !src_info ||
        !layout_range || !status || W <= 0 || W > MI_EP_MAX_RANKS || my_rank < 0 || my_rank >= W || L <= 0 || L * W > 2048 || H <= 0 ||
        H % 16 || max_tokens <= 0 || (epoch == 0 && !epoch_ctr))
        return MI_EP_EINVAL;
    PeerPtrs cp, pp;
    for (int i = 0; i < W; ++i) {
        if (!peer_counts_host[i]) return MI_EP_EINVAL;
        cp.p[i] = peer_counts_host[i];
        pp.p[i] = const_cast<void *>(my_rows);
    }
    hipStream_t s = (hipStream_t)stream;
    const uint64_t ticks = (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull;
    // only ONE workgroup ever spins on the peers (a chip full of spinning workgroups would starve whatever else has to run
    // for the posts to happen when several processes share the GPU); the packing kernel follows on the stream
    const EpochRef er = epoch_ctr ? EpochRef{epoch_ctr, 1} : EpochRef{nullptr, epoch};
    ll_counts_kernel<<<1, 256, (size_t)(L * W + 16) * 4, s>>>(cp, num_tokens_per_expert, my_rank, my_counts, er,
                                                            epoch_ctr ? counts_parity_stride : 0, epoch_ctr, L, W, count_type,
                                                            layout_range, packed_recv_count, status, ticks);
    const int cap = rows_capacity > 0 ? rows_capacity : L * W * max_tokens;      // rows the caller's output buffers hold
    const int payload = H * (quant_mode == MI_EP_QUANT_NONE ? 2 : 1);
    long long blocks = ((long long)L * W * max_tokens + kPullRowsPerBlock - 1) / kPullRowsPerBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    pull_kernel<<<(int)blocks, kWave * kPullWaves, (size_t)L * W * 4, s>>>(pp, layout_range, nullptr, max_tokens, W, L * W, payload,
                                                                         (uint8_t *)packed_recv_x, packed_recv_x_scales, src_info, cap,
                                                                         make_parity(epoch_ctr, 0, rows_parity_stride));
    return launch_status();
}
