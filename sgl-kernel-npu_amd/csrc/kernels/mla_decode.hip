// Paged MLA decode attention for gfx950 (MI355X), bf16 / fp16.
// Replaces the reference Triton kernel _paged_mla_fwd_kernel / decode_mla
// (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:5-230): same math (scores in fp32, online softmax,
// P rounded to the KV dtype, V aliases K_nope, out = acc / l), different schedule.
//
// MI355X design (BASELINE C4: B=128, 128 q-heads sharing one latent KV head, D = 512 + 64, seqlen 4096)
//  * arithmetic intensity ~242 FLOP/B sits under the HBM/MFMA ridge (~400): the KV stream must be read from HBM exactly
//    once.  A workgroup is 4 waves x 16 heads, ONE wave per SIMD so each wave owns the whole 512-entry register file
//    (128 accumulators + 72 resident Q^T registers leave room for deep operand prefetch -- with two waves per SIMD the
//    compiler serialised every ds_read behind its MFMA).  The two 64-head workgroups of a sequence get adjacent
//    workgroup ids on the SAME XCD (id -> XCD is id % 8), so the second reader of a KV tile hits that XCD's L2 and HBM
//    still sees each tile once (the reference launches one program per 16 heads and re-reads KV 8 times);
//  * KV tiles of 64 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR
//    staging), double buffered (2 x 74 KB of the 160 KB LDS), ONE barrier per tile; any page_size works because every
//    key row is addressed through the block table individually;
//  * both GEMMs run transposed so nothing is shuffled between them: S^T[key, head] = K · Q^T (A = K rows from LDS via
//    ds_read_b128, B = Q^T fragments resident in 72 VGPRs) and O^T[d, head] += V^T · P^T (A = V^T through the LDS
//    transpose read ds_read_b64_tr_b16, B = P^T which IS the S^T accumulator layout, packed to bf16 in place);
//    v_mfma_f32_16x16x32_{bf16,f16}; softmax statistics are one value per lane (lane = head);
//  * nope rows are padded to 1056 B (conflict-free ds_read_b128 across the 16 keys of an MFMA operand); the 128-B rope
//    rows are XOR-swizzled by choosing which global chunk each LDS-DMA lane fetches;
//  * flash-decoding split over the KV range fills the 256 CUs when batch < 256 (C4: 2 splits -> 256 workgroups); a small
//    kernel merges the (m, l, O) partials.
// Algorithmic HBM bytes per launch: sum_b len_b * 576 * 2 (KV once) + B*Hq*(576+512)*2 (Q, out); FLOPs
// sum_b Hq * len_b * (576+512) * 2.
#include "device_once.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_sgl_kernels.h"
#include <stdlib.h>

#include "mla_common.h"

// Tile fill: 0 = LDS-DMA (global_load_lds) with one 1-KiB piece issued per QK k-step (default, fastest);
// 1 = register-staged (global_load_dwordx4 -> VGPR -> ds_write_b128), kept as the measured alternative.
#ifndef MLA_STAGE
#define MLA_STAGE 0
#endif

namespace mi_sgl {

// Reductions across the four 16-lane rows of a wave through v_permlane16_swap / v_permlane32_swap (VALU) instead of ds_bpermute (an LDS
// round trip each).  swap16(x): {rows 0,0,2,2 | rows 1,1,3,3} of x; swap32(x): {lower half twice | upper half twice}.  Sums and maxima of
// the two parts are the xor-16 / xor-32 butterfly steps (same two operands in every lane: bit-identical to the shuffle form).
struct RowPair {
    float a, b;
};
__device__ __forceinline__ RowPair swap16(float x)
{
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(x);
    const u32x2v r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return RowPair{__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ RowPair swap32(float x)
{
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(x);
    const u32x2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return RowPair{__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ float max_over_rows(float x)
{
    const RowPair p = swap16(x);
    const RowPair q = swap32(fmaxf(p.a, p.b));
    return fmaxf(q.a, q.b);
}
__device__ __forceinline__ float sum_over_rows(float x)
{
    const RowPair p = swap16(x);
    const RowPair q = swap32(p.a + p.b);
    return q.a + q.b;
}

constexpr int kBufBytes = kTile * kNopeStride + kTile * kRopeStride;   // 74752
#ifndef MLA_QK_AHEAD
#define MLA_QK_AHEAD 3
#endif
#ifndef MLA_WAVES
#define MLA_WAVES 4
#endif
constexpr int kMaxWaves = MLA_WAVES;
// MLA_WAVES == 4: one wave per SIMD, each owns 16 heads x all 512 output dims.
// MLA_WAVES == 8: two waves per SIMD; the pair (w, w+4) shares 16 heads, both compute S (the 72 QK MFMAs are duplicated)
//                 and each accumulates half of the 512 output dims (64 accumulator registers instead of 128), so a wave
//                 fits 256 registers WITH operand prefetch and the SIMD always has a second wave to issue from.
constexpr int kDSplit = kMaxWaves / 4;
constexpr int kHeadWaves = 4;
constexpr bool kDmaInterleaved = !MLA_STAGE && MLA_WAVES == 4;   // 18 DMA pieces per wave per tile = 18 QK k-steps
constexpr int kHeadsPerBlock = kHeadWaves * 16;
constexpr int kAccTiles = 32 / kDSplit;

// LDS-DMA through inline asm: lane l moves 16 B from its own address to dst + 16 l (M0 = wave-uniform LDS destination).  With the
// builtin the compiler put an `s_waitcnt vmcnt(0)` behind EVERY piece (it cannot tell which later ds_read the DMA might feed), i.e. one
// memory round trip per QK k-step; the only wait these requests need is the explicit one at the top of the tile loop.  The compiler's
// own waits (block-table loads, Q^T) stay correct: requests complete in order, so extra ones in flight only make them conservative.
__device__ __forceinline__ void dma16_row(const uint8_t *dst_lds, const void *vaddr)
{
    const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)dst_lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(vaddr) : "memory");
}

// issue the LDS-DMA of one KV tile into `buf`; the 72 wave-instructions are dealt round-robin to the waves
__device__ __forceinline__ void issue_tile(const MlaParams &p, const TileRows &rows, uint8_t *buf, int wave, int nwaves, int lane)
{
    for (int i = wave; i < kTile; i += nwaves) {       // one key row (1 KiB of nope) per instruction; i is wave-uniform
        const int lo = __builtin_amdgcn_readlane((int)(rows.nope & 0xFFFFFFFFll), i);
        const int hi = __builtin_amdgcn_readlane((int)(rows.nope >> 32), i);
        const uint16_t *src = p.k_nope + (((int64_t)hi << 32) | (uint32_t)lo);
        dma16_row(buf + i * kNopeStride, src + lane * 8);
    }
    for (int j = wave; j < kTile / 8; j += nwaves) {   // 8 keys x 128 B of rope per instruction
        const int key = j * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (key & 7);       // XOR swizzle on the source side
        const uint16_t *src = p.k_rope + lane_i64(rows.rope, key);
        dma16_row(buf + kTile * kNopeStride + j * 8 * kRopeStride, src + chunk * 8);
    }
}

// one of the 18 DMA instructions a wave contributes to a tile (4-wave build): pieces 0..15 are nope rows wave + 4*idx,
// pieces 16..17 are the rope blocks wave + 4*(idx - 16).  Lets the caller spread the issue over its MFMA stream.
__device__ __forceinline__ void issue_piece(const MlaParams &p, const TileRows &rows, uint8_t *buf, int wave, int lane, int idx)
{
    if (idx < 16) {
        const int i = wave + 4 * idx;
        const int lo = __builtin_amdgcn_readlane((int)(rows.nope & 0xFFFFFFFFll), i);
        const int hi = __builtin_amdgcn_readlane((int)(rows.nope >> 32), i);
        const uint16_t *src = p.k_nope + (((int64_t)hi << 32) | (uint32_t)lo);
        dma16_row(buf + i * kNopeStride, src + lane * 8);
    } else {
        const int j = wave + 4 * (idx - 16);
        const int key = j * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (key & 7);
        const uint16_t *src = p.k_rope + lane_i64(rows.rope, key);
        dma16_row(buf + kTile * kNopeStride + j * 8 * kRopeStride, src + chunk * 8);
    }
}

// Register-staged tile fill (MLA_STAGE == 1, 4-wave build): global_load_dwordx4 -> VGPR -> ds_write_b128 in two halves of
// 9 x 16 B per thread.  LDS-DMA turned out to be capped near 30 GB/s per CU on this part and to back-pressure the
// issuing (= computing) waves, so fill time added to compute time instead of hiding under it; plain vector loads retire
// asynchronously at the L1 rate (64 B/clk/CU).
struct StageRegs {
    u32x4 v[9];
};

__device__ __forceinline__ void stage_load(const MlaParams &p, const TileRows &rows, int half, int wave, int lane, StageRegs &st)
{
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int i = half * 9 + j;
        if (i < 16) {                                    // nope: row 4i + wave, 16-B column `lane`
            const int key = 4 * i + wave;
            const int lo = __builtin_amdgcn_readlane((int)(rows.nope & 0xFFFFFFFFll), key);
            const int hi = __builtin_amdgcn_readlane((int)(rows.nope >> 32), key);
            st.v[j] = *(const u32x4 *)(p.k_nope + (((int64_t)hi << 32) | (uint32_t)lo) + lane * 8);
        } else {                                         // rope: 8 keys x 8 chunks per wave-instruction
            const int r = (i - 16) * 256 + wave * 64 + lane;
            const int key = r >> 3;
            st.v[j] = *(const u32x4 *)(p.k_rope + lane_i64(rows.rope, key) + (r & 7) * 8);
        }
    }
}

__device__ __forceinline__ void stage_store(uint8_t *buf, int half, int wave, int lane, const StageRegs &st)
{
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int i = half * 9 + j;
        if (i < 16) {
            *(u32x4 *)(buf + (4 * i + wave) * kNopeStride + lane * 16) = st.v[j];
        } else {
            const int r = (i - 16) * 256 + wave * 64 + lane;
            const int key = r >> 3;
            *(u32x4 *)(buf + kTile * kNopeStride + key * kRopeStride + (((r & 7) ^ (key & 7)) * 16)) = st.v[j];
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(64 * kMaxWaves) __attribute__((amdgpu_waves_per_eu(kDSplit, kDSplit))) void mla_decode_kernel(MlaParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = blockDim.x >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    // 1-D grid decode: the head blocks of one (batch, kv head, split) unit sit on one XCD, back to back in dispatch order
    const int head_blocks = (p.group + kHeadsPerBlock - 1) / kHeadsPerBlock;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int unit = (j / head_blocks) * 8 + xcd;
    const int hblk = j % head_blocks;
    // planned form (groups of <= 64 heads, one head block): workgroup = item of the work list, its partial at slot (item, head in group)
    int split, kvh, b, t_begin, t_end, nsp = p.num_splits;
    if (p.plan) {
        const PlanItem it = plan_item(p.plan, (long long)p.batch * p.kv_heads, blockIdx.x);
        if (it.seq < 0) return;
        b = it.seq / p.kv_heads, kvh = it.seq % p.kv_heads, split = it.k, nsp = it.n;
        // the list counts 32-key tiles; pieces start on even ones, except an empty trailing piece clamped to an odd tile count: rounding
        // its start up as well keeps it empty
        constexpr int r = kTile / kWideTile;
        t_begin = (it.t_begin + r - 1) / r, t_end = (it.t_end + r - 1) / r;
    } else {
        if (unit >= p.batch * p.kv_heads * p.num_splits) return;
        split = unit % p.num_splits;
        kvh = (unit / p.num_splits) % p.kv_heads;
        b = unit / (p.num_splits * p.kv_heads);
    }
    const int seq_len = p.seq_lens[b];
    if (p.plan) {      // pieces clamped to the tiles the sequence has NOW, the last one running to their end (a stale list costs balance only)
        const int ntiles = (seq_len + kTile - 1) / kTile;
        t_begin = min(t_begin, ntiles);
        t_end = split == nsp - 1 ? ntiles : min(t_end, ntiles);
    }
    if (!p.plan) {
        const int ntiles = (seq_len + kTile - 1) / kTile;
        const int tps = (ntiles + p.num_splits - 1) / p.num_splits;
        t_begin = split * tps;
        t_end = min(ntiles, t_begin + tps);
    }
    const int dhalf = wave / kHeadWaves;                         // which slice of the 512 output dims this wave accumulates
    const int hg = hblk * kHeadsPerBlock + (wave % kHeadWaves) * 16 + c16;      // head inside the kv group
    const bool head_ok = hg < p.group;
    const int head = kvh * p.group + hg;

    // Q^T fragments: lane (g, c16) holds q[head][ks*32 + g*8 .. +8]
    s16x8 qf[18];
    {
        const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            if (head_ok) qf[ks] = *(const s16x8 *)(qrow + ks * 32 + g * 8);
            else qf[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x4 acc[kAccTiles];
#pragma unroll
    for (int i = 0; i < kAccTiles; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    TileRowsRaw raw_next = tile_rows_load(p, b, seq_len, t_begin, lane);
    TileRows rows_nxt{0, 0};
    StageRegs st, st2;
    if (t_begin < t_end) {
        if (MLA_STAGE) {
            const TileRows r0 = tile_rows_finish(p, kvh, raw_next);
            stage_load(p, r0, 0, wave, lane, st);
            stage_store(lds, 0, wave, lane, st);
            stage_load(p, r0, 1, wave, lane, st);
            stage_store(lds, 1, wave, lane, st);
        } else {
            issue_tile(p, tile_rows_finish(p, kvh, raw_next), lds, wave, nwaves, lane);
        }
        raw_next = tile_rows_load(p, b, seq_len, t_begin + 1, lane);
    }
    for (int t = t_begin; t < t_end; ++t) {
        uint8_t *buf = lds + ((t - t_begin) & 1) * kBufBytes;
        uint8_t *nbuf = lds + ((t + 1 - t_begin) & 1) * kBufBytes;
        const bool more = t + 1 < t_end;
        if (!MLA_STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces of tile t have landed
        __syncthreads();                                      // tile t is complete in LDS, tile t-1 is no longer read
        if (MLA_STAGE) {
            if (more) {
                rows_nxt = tile_rows_finish(p, kvh, raw_next);
                stage_load(p, rows_nxt, 0, wave, lane, st);   // first half of tile t+1 travels under the QK MFMAs
                raw_next = tile_rows_load(p, b, seq_len, t + 2, lane);
            }
        } else if (kDmaInterleaved) {
            rows_nxt = tile_rows_finish(p, kvh, raw_next);     // past the last tile the rows are clamped: a harmless dummy fill
        } else if (more) {
            issue_tile(p, tile_rows_finish(p, kvh, raw_next), nbuf, wave, nwaves, lane);
            raw_next = tile_rows_load(p, b, seq_len, t + 2, lane);      // consumed one iteration later, after the wait above
        }

        // ---- S^T[key, head] = K · Q^T  (4 m-tiles of 16 keys, 18 k-steps of 32 dims)
        f32x4 s[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto lda = [&](int ks, int mt) -> s16x8 {
            if (ks < 16) return *(const s16x8 *)(buf + (mt * 16 + c16) * kNopeStride + ks * 64 + g * 16);
            const int key = mt * 16 + c16;
            const int chunk = ((ks - 16) * 4 + g) ^ (key & 7);
            return *(const s16x8 *)(buf + kTile * kNopeStride + key * kRopeStride + chunk * 16);
        };
        // Explicit software pipeline, fenced with sched_barrier so the machine scheduler cannot sink the prefetch back
        // next to its use: operands of k-step ks + kAhead are issued before the MFMAs of k-step ks.
        constexpr int kAhead = MLA_QK_AHEAD;
        s16x8 af[kAhead + 1][4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pre = 0; pre < kAhead; ++pre)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[pre][mt] = lda(pre, mt);
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (ks + kAhead < 18) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) af[(ks + kAhead) % (kAhead + 1)][mt] = lda(ks + kAhead, mt);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) s[mt] = mfma16<BF16>(af[ks % (kAhead + 1)][mt], qf[ks], s[mt]);
            if (kDmaInterleaved) issue_piece(p, rows_nxt, nbuf, wave, lane, ks);    // tile t+1 trickles in under the MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kDmaInterleaved) raw_next = tile_rows_load(p, b, seq_len, t + 2, lane);
        if (MLA_STAGE && more) {
            stage_load(p, rows_nxt, 1, wave, lane, st2);      // second half travels under softmax + PV
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- online softmax; lane owns head c16 and keys mt*16 + 4g + r.  Scores stay unscaled: with
        // c = sm_scale * log2(e), P = exp2(S*c - m*c) is one FMA + one v_exp_f32 per element; m_run / tmax are kept in
        // the scaled log2 domain.  Only the tile that crosses seq_len needs the key mask.
        const float cs = p.sm_scale * 1.4426950408889634f;
        if ((t + 1) * kTile > seq_len) {
            const int kbase = t * kTile + 4 * g;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kbase + mt * 16 + r >= seq_len) s[mt][r] = -INFINITY;
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) tmax = fmaxf(fmaxf(tmax, fmaxf(s[mt][0], s[mt][1])), fmaxf(s[mt][2], s[mt][3]));
        tmax = max_over_rows(tmax);
        tmax *= cs;                                     // sm_scale > 0: max commutes with the scaling
        // Deferred rescale: the running reference m_run only moves when some head's tile maximum exceeds it by more
        // than kDefer (then every head of the wave re-bases to its true maximum); otherwise P = exp2(.) <= 2^kDefer
        // and neither l nor the 128 accumulators need touching.  out = acc / l is invariant to the reference.
        constexpr float kDefer = 11.0f;                 // log2 domain (e^8 ~ 2^11.5)
        if (__any(tmax > m_run + kDefer)) {
            const float m_new = fmaxf(m_run, tmax);
            const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);   // m_run = -inf -> 0
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int c = 0; c < kAccTiles / 8; ++c) {           // 8 accumulators at a time keeps the VGPR<->AGPR staging small
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[c * 8 + i] *= alpha;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float nm = (m_run == -INFINITY) ? 0.f : -m_run;
        float psum = 0.f;
        uint32_t pk[8];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[mt][r], cs, nm));
                psum += e[r];
            }
            pk[mt * 2 + 0] = pack2<BF16>(e[0], e[1]);
            pk[mt * 2 + 1] = pack2<BF16>(e[2], e[3]);
        }
        psum = sum_over_rows(psum);
        l_run += psum;
        if (MLA_STAGE && more) {
            __builtin_amdgcn_sched_barrier(0);
            stage_store(nbuf, 0, wave, lane, st);             // first half had QK + softmax to arrive
            __builtin_amdgcn_sched_barrier(0);
        }
        // P^T fragments: k-step kk covers key tiles (2kk, 2kk+1); slots 0..3 / 4..7 of lane group g
        s16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4 w = u32x4{pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
            pf[kk] = __builtin_bit_cast(s16x8, w);
        }
        // ---- O^T[d, head] += V^T · P^T   (V = nope part of the same LDS tile, transposed on the fly)
        const uint8_t *vrow = buf + (4 * g + (c16 >> 2)) * kNopeStride + (c16 & 3) * 8 + dhalf * kAccTiles * 32;
        auto ldv = [&](int i, int half) -> s16x4 {      // i = kk * kAccTiles + dt
            const int kk = i / kAccTiles, dt = i % kAccTiles;
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (s16x4 __attribute__((address_space(3))) *)(vrow + (2 * kk + half) * 16 * kNopeStride + dt * 32));
        };
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2 * kAccTiles; ++i) {
            const s16x4 lo = ldv(i, 0), hi = ldv(i, 1);
            const s16x8 a = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            acc[i % kAccTiles] = mfma16<BF16>(a, pf[i / kAccTiles], acc[i % kAccTiles]);
        }
        constexpr int kPvAhead = kDSplit == 1 ? 8 : 6;          // MFMAs' worth of tr-reads in flight (<= 15 DS ops)
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * kPvAhead, 0);
#pragma unroll
        for (int i = 0; i < 2 * kAccTiles - kPvAhead; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kPvAhead, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MLA_STAGE && more) stage_store(nbuf, 1, wave, lane, st2);
    }

    if (kDmaInterleaved) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the dummy fill issued under the last tile
    // ---- epilogue: lane holds O^T[d = dt*16 + 4g + r][head c16]
    if (!head_ok) return;
    if (nsp == 1) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)head * p.o_sh + 4 * g + dhalf * kAccTiles * 16;
#pragma unroll
        for (int dt = 0; dt < kAccTiles; ++dt) {
            const uint32_t w0 = (uint32_t)cvt_out<BF16>(acc[dt][0] * inv) | ((uint32_t)cvt_out<BF16>(acc[dt][1] * inv) << 16);
            const uint32_t w1 = (uint32_t)cvt_out<BF16>(acc[dt][2] * inv) | ((uint32_t)cvt_out<BF16>(acc[dt][3] * inv) << 16);
            *(uint2 *)(orow + dt * 16) = uint2{w0, w1};
        }
    } else {
        const int64_t idx = p.plan ? (int64_t)blockIdx.x * p.group + hg : ((int64_t)b * p.q_heads + head) * p.num_splits + split;
        float *po = p.ws_o + idx * kDN + 4 * g + dhalf * kAccTiles * 16;
#pragma unroll
        for (int dt = 0; dt < kAccTiles; ++dt) *(f32x4 *)(po + dt * 16) = acc[dt];
        if (g == 0 && dhalf == 0) {
            p.ws_ml[idx * 2 + 0] = m_run;
            p.ws_ml[idx * 2 + 1] = l_run;
        }
    }
}

// merge the flash-decoding partials: one wave per (b, head); lane handles 8 of the 512 dims.  With fix_only set (wide
// kernel launches) it first checks the sequence's hand-off word and takes the slow path above instead.
template <bool BF16>
__global__ __launch_bounds__(256) void mla_merge_kernel(MlaParams p)
{
    const int lane = threadIdx.x & 63;
    const int64_t bh = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (bh >= (int64_t)p.batch * p.q_heads) return;
    const int b = (int)(bh / p.q_heads), h = (int)(bh % p.q_heads);
    // the meeting words of the sequence's two pieces (mla_decode_wide8s.hip) are re-armed here, behind the launch that used them: the
    // next call -- or the next replay of a captured one, which carries the same tag -- finds them clear
    if (p.pair_flags && h % p.group == 0 && lane < 2) p.pair_flags[((int64_t)b * p.kv_heads + h / p.group) * 2 + lane] = 0ull;
    // nothing left for this launch (every sequence finished in the kernel: BASELINE C4's two pieces per sequence, or one piece each):
    // one load per wave instead of the list lookup and the statistics round trips
    if (p.need_merge && __hip_atomic_load(p.need_merge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.pair_tag) return;
    if (p.fix_only && p.fix_flags[b * p.kv_heads + h / p.group] == p.fix_epoch) {
        mla_recompute_head<BF16>(p, b, h, lane);
        return;
    }
    // partial s of this head: slot (bh, s) of the uniform form, or -- planned form -- slot (first item of the sequence + s, head within the group)
    int S = p.num_splits, first = 0;
    const int hg = h % p.group;
    if (p.plan) {
        const int32_t *info = p.plan + kPlanHdr + 2ll * (b * p.kv_heads + h / p.group);
        first = info[0], S = info[1];
    }
    S = __builtin_amdgcn_readfirstlane(S);                  // the same for every lane of the wave: one (sequence, head) per wave
    if (S == 1) return;
    auto slot = [&](int s) -> int64_t { return p.plan ? (int64_t)(first + s) * p.group + hg : bh * S + s; };
    float L = 0.f;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    if (S <= 64) {
        // One piece per LANE for the bookkeeping: lane s fetches piece s's slot (planned form: one load of base[s]), its maximum and its
        // sum in parallel; the rows are then added in piece order -- the same products and the same order as the plain loop below, whose
        // two dependent loads per piece and pass (base[s], then the word behind it) cost ~1 us per piece: a sequence cut into 64 pieces
        // merged in 140 us, now in 8.
        const int64_t my = lane < S ? slot(lane) : 0;
        const float m = lane < S ? p.ws_ml[my * 2] : -INFINITY;
        float l = lane < S ? p.ws_ml[my * 2 + 1] : 0.f;
        // two pieces: piece hg / 64 finished this head itself (mla_decode_wide8s.hip) if its sum carries the mark (the sign); without the
        // mark (the partner did not show up in time) both partials of the head are complete in the workspace: merge as usual
        if (p.pair_flags && S == 2 && ((__ballot(__float_as_uint(l) >> 31) >> (hg >> 6)) & 1ull)) return;
        l = fabsf(l);
        float M = m;
        for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
        const float w = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - M);      // m is kept in the scaled log2 domain
        const int my_lo = (int)(my & 0xFFFFFFFFll), my_hi = (int)(my >> 32);
        for (int s0 = 0; s0 < S; s0 += 4) {
            f32x4 a[4], c[4];
            float ws[4], ls[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                   // four rows requested together (wave-uniform conditions)
                const int s = min(s0 + u, S - 1);
                ws[u] = s0 + u < S ? __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), s)) : 0.f;
                ls[u] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(l), s));
                const int64_t sl = ((int64_t)__builtin_amdgcn_readlane(my_hi, s) << 32) | (uint32_t)__builtin_amdgcn_readlane(my_lo, s);
                const float *po = p.ws_o + sl * kDN + lane * 8;
                if (ws[u] != 0.f) a[u] = *(const f32x4 *)po, c[u] = *(const f32x4 *)(po + 4);
                else a[u] = c[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (ws[u] == 0.f) continue;                 // an empty piece (m = -inf), or behind the last one
                L += ws[u] * ls[u];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] += ws[u] * a[u][j];
                    o[4 + j] += ws[u] * c[u][j];
                }
            }
        }
    } else {
        float M = -INFINITY;
        for (int s = 0; s < S; ++s) M = fmaxf(M, p.ws_ml[slot(s) * 2]);
        for (int s = 0; s < S; ++s) {
            const float *ml = p.ws_ml + slot(s) * 2;
            const float m = ml[0];
            if (m == -INFINITY) continue;
            const float w = __builtin_amdgcn_exp2f(m - M);
            L += w * ml[1];
            const float *po = p.ws_o + slot(s) * kDN + lane * 8;
            const f32x4 a = *(const f32x4 *)po, c = *(const f32x4 *)(po + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] += w * a[j];
                o[4 + j] += w * c[j];
            }
        }
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh + lane * 8;
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        w[j] = (uint32_t)cvt_out<BF16>(o[2 * j] * inv) | ((uint32_t)cvt_out<BF16>(o[2 * j + 1] * inv) << 16);
    *(u32x4 *)orow = w;
}

}  // namespace mi_sgl

using namespace mi_sgl;

extern "C" const char *mi_sgl_kernels_version(void) { return "mi_sgl_kernels 0.1 gfx950"; }

// workspace = [flash-decoding partials (num_splits > 1)] [batch * q_heads words: one "recompute" flag per (sequence, kv head) for the
// wide kernel's hand-off, then one meeting word per (sequence, kv head, 128-head block) for its in-kernel merge]
static size_t partial_bytes(int batch, int q_heads, int num_splits)
{
    return num_splits <= 1 ? 0 : (size_t)batch * q_heads * num_splits * (kDN + 2) * sizeof(float);
}

// workgroups the chip runs at once with one wide workgroup per CU (the plan's `workers`); 256 when no device answers
static int plan_workers()
{
    static int cached[64];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 256;
    if (!cached[d]) {
        int n = 0;
        cached[d] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[d];
}
// planned form: one partial row per (item, head of its kv group).  kv_heads is not an argument of the workspace query; for any kv_heads
// the rows are items * group <= (seqs + workers + pad) * group = batch * q_heads + (workers + pad) * group, and the planned form serves
// groups of <= 128 heads; the list itself is sized for the most (sequence, kv head) pairs the query allows, batch * q_heads
static long long planned_seqs_bound(int batch, int q_heads) { return (long long)batch * q_heads; }
static long long planned_rows_cap(int batch, int q_heads, int workers)
{
    return (long long)batch * q_heads + (plan_items_max(0, workers)) * (long long)(q_heads < 128 ? q_heads : 128);
}
static size_t planned_partial_bytes(int batch, int q_heads, int workers) { return (size_t)planned_rows_cap(batch, q_heads, workers) * (kDN + 2) * sizeof(float); }

extern "C" size_t mi_mla_decode_workspace(int batch, int q_heads, int num_splits)
{
    if (batch <= 0 || q_heads <= 0) return 0;
    if (num_splits == MI_MLA_SPLITS_PLANNED) {
        const long long seqs = planned_seqs_bound(batch, q_heads);
        const int workers = plan_workers();
        return planned_partial_bytes(batch, q_heads, workers) + (size_t)batch * q_heads * sizeof(uint32_t) + plan_words(seqs, workers) * sizeof(int32_t);
    }
    return partial_bytes(batch, q_heads, num_splits) + (size_t)batch * q_heads * sizeof(uint32_t);   // q_heads >= kv_heads
}

static bool use_wide(int group)
{
    // MI_MLA_WIDE=0: groups of more than 64 heads also run the 64-head kernel, as ceil(group / 64) sibling workgroups per sequence that
    // share an XCD (A/B switch; DESIGN section 4.1 has the measurement)
    static const bool allow = !(getenv("MI_MLA_WIDE") && atoi(getenv("MI_MLA_WIDE")) == 0);
    return allow && group > kHeadsPerBlock && MLA_WAVES == 4;
}

// introspection for tests and tuning: where the work list sits in a planned workspace, and the worker count it was built for
extern "C" size_t mi_mla_decode_plan_offset(int batch, int q_heads)
{
    if (batch <= 0 || q_heads <= 0) return 0;
    return planned_partial_bytes(batch, q_heads, plan_workers()) + (size_t)batch * q_heads * sizeof(uint32_t);
}
extern "C" int mi_mla_decode_plan_workers(void) { return plan_workers(); }

static int g_wide_variant = 0;      // 0 = environment / default; 4, 8 or 9 = forced (tests run every form in one process)
static int wide_variant()
{
    // MI_MLA_WIDE8 = 0: four waves; 1: eight waves, per-key block ids through an LDS ring (mla_decode_wide8.hip; any page size); 2 (default):
    // eight waves, one scalar block id per tile and three tiles in flight (mla_decode_wide8s.hip; power-of-two pages of >= 32 keys, other
    // page sizes take the ring kernel)
    static const int wide_env = getenv("MI_MLA_WIDE8") ? (atoi(getenv("MI_MLA_WIDE8")) == 0 ? 4 : atoi(getenv("MI_MLA_WIDE8")) == 1 ? 8 : 9) : 9;
    return g_wide_variant ? g_wide_variant : wide_env;
}
static bool wide8_selected() { return wide_variant() >= 8; }
// the planned form serves one workgroup per (sequence, kv head) piece: the eight-wave wide kernel with one 128-head block, or the 64-head
// kernel with one head block (groups of <= 64 heads: the 16-head shards of a TP-8 deployment); MI_MLA_PLAN=0 keeps uniform splits
static bool plan_applies(int group)
{
    static const bool allow = !(getenv("MI_MLA_PLAN") && atoi(getenv("MI_MLA_PLAN")) == 0);
    if (!allow) return false;
    return use_wide(group) ? (group <= 128 && wide8_selected()) : group <= kHeadsPerBlock;
}

static int uniform_splits(int batch, int q_heads, int kv_heads, int max_seq_len);
extern "C" int mi_mla_decode_num_splits(int batch, int q_heads, int kv_heads, int max_seq_len)
{
    if (batch <= 0 || q_heads <= 0 || kv_heads <= 0 || max_seq_len <= 0) return 1;
    if (plan_applies(q_heads / kv_heads)) return MI_MLA_SPLITS_PLANNED;
    return uniform_splits(batch, q_heads, kv_heads, max_seq_len);
}
// positive uniform split count (what mi_mla_decode_num_splits returned before the planned form existed): for callers that size their
// own loops / workspaces by it
extern "C" int mi_mla_decode_uniform_splits(int batch, int q_heads, int kv_heads, int max_seq_len)
{
    if (batch <= 0 || q_heads <= 0 || kv_heads <= 0 || max_seq_len <= 0) return 1;
    return uniform_splits(batch, q_heads, kv_heads, max_seq_len);
}
static int uniform_splits(int batch, int q_heads, int kv_heads, int max_seq_len)
{
    const int group = q_heads / kv_heads;
    const int hpb = use_wide(group) ? 128 : kHeadsPerBlock, tile = use_wide(group) ? kWideTile : kTile;
    const long long wgs = (long long)batch * kv_heads * ((group + hpb - 1) / hpb);
    const int ntiles = (max_seq_len + tile - 1) / tile;
    int s = (int)((256 + wgs - 1) / wgs);          // at least one workgroup per CU
    const int min_tiles = 256 / tile;                  // keep >= 256 keys per split
    const int cap = ntiles / min_tiles > 1 ? ntiles / min_tiles : 1;
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

static int g_pair_mode = -1;        // -1 = environment / default (on); 0 = off, 1 = on, 2 = on with piece 1 withholding its word (tests)
extern "C" int mi_mla_decode_set_pair(int mode)
{
    if (mode < -1 || mode > 2) return MI_SGL_EINVAL;
    g_pair_mode = mode;
    return MI_SGL_OK;
}

extern "C" int mi_mla_decode_select_wide(int waves)
{
    if (waves != 0 && waves != 4 && waves != 8 && waves != 9) return MI_SGL_EINVAL;
    g_wide_variant = waves;
    return MI_SGL_OK;
}

// A work list for mi_mla_decode_with_plan: built once from kv_seq_lens (one small launch), valid for every call on the same kv_seq_lens
// CONTENTS, batch and kv_heads -- the 61 attention layers of a decode step share one.
extern "C" size_t mi_mla_decode_plan_bytes(int batch, int kv_heads)
{
    if (batch <= 0 || kv_heads <= 0) return 0;
    return plan_words((long long)batch * kv_heads, plan_workers()) * sizeof(int32_t);
}
extern "C" int mi_mla_decode_build_plan(const int32_t *kv_seq_lens, int batch, int kv_heads, void *plan, size_t plan_bytes, void *stream)
{
    if (batch <= 0 || kv_heads <= 0 || !kv_seq_lens || !plan || plan_bytes < mi_mla_decode_plan_bytes(batch, kv_heads)) return MI_SGL_EINVAL;
    decode_plan_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(kv_seq_lens, batch, kv_heads, kWideTile, kTile / kWideTile, plan_workers(), (int32_t *)plan);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

static int mla_decode_impl(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                           const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                           int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk,
                           int64_t kn_stride_row, int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row,
                           int64_t kr_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype,
                           int num_splits, void *workspace, size_t workspace_bytes, void *stream, const int32_t *ready_plan);
extern "C" int mi_mla_decode(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                             const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                             int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk,
                             int64_t kn_stride_row, int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row,
                             int64_t kr_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype,
                             int num_splits, void *workspace, size_t workspace_bytes, void *stream)
{
    return mla_decode_impl(q, k_nope, k_rope, out, kv_seq_lens, block_table, batch, q_heads, kv_heads, page_size, bt_stride, max_seq_len, q_stride_b,
                           q_stride_h, kn_stride_blk, kn_stride_row, kn_stride_h, kr_stride_blk, kr_stride_row, kr_stride_h, o_stride_b, o_stride_h,
                           sm_scale, dtype, num_splits, workspace, workspace_bytes, stream, nullptr);
}
// mi_mla_decode in the planned form with a work list the caller built earlier (mi_mla_decode_build_plan): no plan launch in front of the
// kernel.  workspace as for num_splits = MI_MLA_SPLITS_PLANNED.  Returns MI_SGL_ENOTAPPLICABLE where the planned form does not apply
// (kv groups of more than 128 heads; for groups of 65..128: page sizes that are not powers of two, the four-wave kernel selected): call
// mi_mla_decode.
extern "C" int mi_mla_decode_with_plan(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                                       const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                                       int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk,
                                       int64_t kn_stride_row, int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row,
                                       int64_t kr_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype,
                                       const void *plan, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!plan) return MI_SGL_EINVAL;
    return mla_decode_impl(q, k_nope, k_rope, out, kv_seq_lens, block_table, batch, q_heads, kv_heads, page_size, bt_stride, max_seq_len, q_stride_b,
                           q_stride_h, kn_stride_blk, kn_stride_row, kn_stride_h, kr_stride_blk, kr_stride_row, kr_stride_h, o_stride_b, o_stride_h,
                           sm_scale, dtype, MI_MLA_SPLITS_PLANNED, workspace, workspace_bytes, stream, (const int32_t *)plan);
}

static int mla_decode_impl(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                           const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                           int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk,
                           int64_t kn_stride_row, int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row,
                           int64_t kr_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype,
                           int num_splits, void *workspace, size_t workspace_bytes, void *stream, const int32_t *ready_plan)
{
    if (batch < 0 || q_heads <= 0 || kv_heads <= 0 || q_heads % kv_heads || page_size <= 0 || bt_stride <= 0) return MI_SGL_EINVAL;
    if (batch == 0) return MI_SGL_OK;
    if (!q || !k_nope || !k_rope || !out || !kv_seq_lens || !block_table) return MI_SGL_EINVAL;
    if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_SGL_EINVAL;
    if ((q_stride_h % 8) || (q_stride_b % 8) || (kn_stride_row % 8) || (kn_stride_blk % 8) || (kn_stride_h % 8) ||
        (kr_stride_row % 8) || (kr_stride_blk % 8) || (kr_stride_h % 8) || (o_stride_h % 8) || (o_stride_b % 8))
        return MI_SGL_EINVAL;      // 16-byte vector accesses
    if (num_splits == 0) num_splits = mi_mla_decode_num_splits(batch, q_heads, kv_heads, max_seq_len);
    if (num_splits < 0 && num_splits != MI_MLA_SPLITS_PLANNED) return MI_SGL_EINVAL;
    // the wide kernel addresses the cache with 32-bit block strides and 24-bit row strides (elements); anything else (blocks of 4 GB,
    // rows of 32 MB) goes to the 64-head kernel, which keeps the general int64 form
    auto fits = [](int64_t v, int bits) { return v >= 0 && v < (1ll << bits); };
    const bool narrow = fits(kn_stride_blk, 31) && fits(kr_stride_blk, 31) && fits(kn_stride_row, 24) && fits(kr_stride_row, 24) &&
                        page_size < (1 << 24) &&
                        // row-in-block * row stride is formed with a 24 x 24 -> 32-bit multiply: the product must fit 32 bits too
                        fits((int64_t)(page_size - 1) * kn_stride_row, 32) && fits((int64_t)(page_size - 1) * kr_stride_row, 32);
    const bool wide = use_wide(q_heads / kv_heads) && narrow;
    // planned form asked for but not served by this launch (four-wave kernel selected, > 128 heads per group, a page size that is not a
    // power of two, cache strides that need the 64-head kernel): uniform splits, as many as the caller's workspace holds
    bool planned = num_splits == MI_MLA_SPLITS_PLANNED;
    const int group = q_heads / kv_heads;
    const bool plan_ok = plan_applies(group) && (use_wide(group) ? wide && (page_size & (page_size - 1)) == 0 : true);
    if (planned && !plan_ok) {
        if (ready_plan) return MI_SGL_ENOTAPPLICABLE;
        planned = false;
        num_splits = uniform_splits(batch, q_heads, kv_heads, max_seq_len);
        while (num_splits > 1 && mi_mla_decode_workspace(batch, q_heads, num_splits) > workspace_bytes) --num_splits;
    }
    if ((num_splits > 1 || wide) && (!workspace || workspace_bytes < mi_mla_decode_workspace(batch, q_heads, planned ? MI_MLA_SPLITS_PLANNED : num_splits)))
        return MI_SGL_EINVAL;
    MlaParams p;
    p.plan = nullptr;
    const long long seqs = (long long)batch * kv_heads;
    const int workers = plan_workers();
    p.q = (const uint16_t *)q, p.k_nope = (const uint16_t *)k_nope, p.k_rope = (const uint16_t *)k_rope;
    p.out = (uint16_t *)out, p.seq_lens = kv_seq_lens, p.block_table = block_table;
    p.ws_o = (float *)workspace;
    p.ws_ml = p.ws_o ? p.ws_o + (planned ? (size_t)planned_rows_cap(batch, q_heads, workers) : (size_t)batch * q_heads * num_splits) * kDN : nullptr;
    static uint32_t epoch = 0;
    // (planned: the partial area is sized for the workspace query's bound on the sequences, >= this call's)
    const size_t part_bytes = planned ? planned_partial_bytes(batch, q_heads, workers) : partial_bytes(batch, q_heads, num_splits);
    p.fix_flags = workspace ? (uint32_t *)((char *)workspace + part_bytes) : nullptr;
    if (planned) {
        int32_t *plan = (int32_t *)((char *)workspace + part_bytes + (size_t)batch * q_heads * sizeof(uint32_t));
        if (!ready_plan) decode_plan_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(kv_seq_lens, batch, kv_heads, kWideTile, kTile / kWideTile, workers, plan);
        p.plan = ready_plan ? ready_plan : plan;
        num_splits = 1;                                // per sequence now: the kernels read it from the list
    }
    p.fix_epoch = ++epoch ? epoch : ++epoch;          // a stale word equal to the epoch only causes a redundant recompute
    p.fix_only = 0;
    p.arrive = p.fix_flags ? p.fix_flags + (size_t)batch * kv_heads : nullptr;       // inside the batch * q_heads flag words (wide: group > 64)
    p.pair_flags = nullptr, p.pair_tag = 0, p.pair_withhold = 0, p.need_merge = nullptr;
    // The wide kernel finishes a one-split launch itself (slow path included): no second launch.  With two splits its
    // in-kernel merge is correct (MI_MLA_INLINE_MERGE=2 enables it) but measured 202 us against 195 us for the separate
    // merge launch at C4: the agent-scope release / acquire each pair needs costs an L2 write-back and an L2 invalidate.
    static const int inline_splits = getenv("MI_MLA_INLINE_MERGE") ? atoi(getenv("MI_MLA_INLINE_MERGE")) : 1;
    // kv groups of more than 64 heads: the eight-wave wide kernel (two waves per SIMD; default) or the four-wave one (one per SIMD;
    // MI_MLA_WIDE8=0 or mi_mla_decode_select_wide(4)) -- same numerics contract, both under the whole test matrix; alternating in one
    // process at BASELINE C4 the eight-wave form is 1 % faster on full sequences (178.5 vs 180.5 us with the merge) and 4 % on ragged ones
    // (135.9 vs 141.4 us), DESIGN section 4.1
    const bool wide8 = wide8_selected();
    p.inline_merge = !planned && wide && (wide8 ? num_splits == 1 : (num_splits <= inline_splits && num_splits <= 2));
    p.batch = batch, p.q_heads = q_heads, p.kv_heads = kv_heads, p.group = q_heads / kv_heads, p.page_size = page_size;
    p.bt_stride = bt_stride, p.num_splits = num_splits;
    p.q_sb = q_stride_b, p.q_sh = q_stride_h, p.kn_sblk = kn_stride_blk, p.kn_srow = kn_stride_row, p.kn_sh = kn_stride_h;
    p.kr_sblk = kr_stride_blk, p.kr_srow = kr_stride_row, p.kr_sh = kr_stride_h, p.o_sb = o_stride_b, p.o_sh = o_stride_h;
    p.sm_scale = sm_scale;
    hipStream_t st = (hipStream_t)stream;
    const long long units = (long long)batch * kv_heads * num_splits;
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        const int lds1 = 2 * kBufBytes;
        (void)hipFuncSetAttribute((const void *)mla_decode_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
        (void)hipFuncSetAttribute((const void *)mla_decode_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    }
    if (wide) {
        const bool scalar_ids = wide_variant() == 9 && (page_size & (page_size - 1)) == 0 && page_size >= kWideTile;
        // sequences in two pieces finish between their two workgroups (mla_decode_wide8s.hip; MI_MLA_PAIR=0: through the merge kernel
        // like every other split count).  The meeting words live in the flag area behind the per-sequence recompute words: groups of
        // 65..128 heads leave >= 64 words per sequence there, the pair takes four (+ one for alignment).
        static const bool pair_env = !(getenv("MI_MLA_PAIR") && atoi(getenv("MI_MLA_PAIR")) == 0);
        const bool pair_on = g_pair_mode < 0 ? pair_env : g_pair_mode != 0;
        p.pair_withhold = g_pair_mode == 2;
        if (pair_on && scalar_ids && wide8 && p.group <= 128 && (planned || num_splits == 2) && p.arrive) {
            p.pair_flags = (uint64_t *)(((uintptr_t)p.arrive + 7) & ~(uintptr_t)7);
            p.need_merge = p.pair_flags + 2 * (size_t)batch * kv_heads;
            p.pair_tag = ((uint64_t)p.fix_epoch * 0x9E3779B97F4A7C15ull) | 8ull;      // never 0 above the three bits that carry the XCC id
        }
        if (planned) (scalar_ids ? launch_mla_wide8s : launch_mla_wide8)(p, dtype, plan_items_max(seqs, workers), st);
        else if (wide8) (scalar_ids ? launch_mla_wide8s : launch_mla_wide8)(p, dtype, units, st);
        else launch_mla_wide(p, dtype, units, st);
        p.fix_only = 1;                                // the merge kernel also serves as the slow path for flagged sequences
    } else {
        const int head_blocks = (p.group + kHeadsPerBlock - 1) / kHeadsPerBlock;
        const int nwaves = kHeadWaves * kDSplit;      // head waves beyond the group size only help with the DMA
        dim3 grid((unsigned)(planned ? plan_items_max(seqs, workers) : ((units + 7) / 8) * 8 * head_blocks));
        const size_t lds = 2 * (size_t)kBufBytes;
        if (dtype == MI_DTYPE_BF16) mla_decode_kernel<true><<<grid, 64 * nwaves, lds, st>>>(p);
        else mla_decode_kernel<false><<<grid, 64 * nwaves, lds, st>>>(p);
    }
    if ((num_splits > 1 || wide || planned) && !p.inline_merge) {
        const long long bh = (long long)batch * q_heads;
        const int blocks = (int)((bh + 3) / 4);
        if (dtype == MI_DTYPE_BF16) mla_merge_kernel<true><<<blocks, 256, 0, st>>>(p);
        else mla_merge_kernel<false><<<blocks, 256, 0, st>>>(p);
    }
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}
