// Paged grouped-query decode attention with separate K / V caches for gfx950 -- the generic sibling of mla_decode.hip.
// Replaces the reference's Triton-Ascend `_paged_gqa_fwd_kernel` / `decode_gqa` and `decode_gqa_high_performance`
// (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:233-450, :646-760): one query token per sequence,
// q [B, Hq, Lk], K cache [blocks, page, Hkv, Lk], V cache [blocks, page, Hkv, Lv], fp32 scores / online softmax, P rounded
// to the cache dtype before P.V (`p_exp.to(v.dtype)`, :362).
//
// MI355X design (HBM-bound: every K/V byte is used once per kv head): a workgroup of 4 waves owns one (sequence, kv head,
// KV split) and ALL query heads of the group (wave w takes head blocks w, w+4, ...; 16 heads = the N dimension of the MFMA),
// so K and V leave HBM once -- the reference re-reads them per 32-head block.  Tiles of 64 (or 32) keys are staged
// global -> VGPR -> LDS (8 threads per key row = 128 contiguous bytes per row per instruction), double buffered, one
// barrier per tile; the next tile's loads are in flight during the current tile's MFMAs.  Both GEMMs are transposed as in
// the MLA kernel: S^T = K.Q^T (A = K rows, ds_read_b128; B = Q^T resident in registers), O^T += V^T.P^T (A = V^T through
// ds_read_b64_tr_b16, B = P^T = the S^T accumulator layout packed in place) -- no cross-lane traffic between the GEMMs.
// LDS rows are padded to an odd number of 32-byte units, which makes both read patterns conflict-free.
// Flash-decoding split over the KV range + a merge kernel fill the 256 CUs for small batch x kv_heads.
#include <algorithm>

#include "device_once.h"
#include "decode_plan.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "gqa_wide.h"
#include "mi_sgl_kernels.h"

namespace mi_gqa {

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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct GqaParams {
    const uint16_t *q, *k, *v;
    uint16_t *out;
    const int32_t *seq_lens, *block_table;
    float *ws_o;      // [B][Hq][S][DVP] fp32 partial (unnormalised) outputs
    float *ws_ml;     // [B][Hq][S][2]   running max (scaled log2 domain), running sum
    int batch, q_heads, kv_heads, group, page_size, bt_stride, num_splits, lk, lv;
    int64_t q_sb, q_sh, k_sblk, k_srow, k_sh, v_sblk, v_srow, v_sh, o_sb, o_sh;
    float sm_scale;
    // attention with sinks (attention/sinks_attention.py:7-137, :139-286): a per-q-head logit that takes part in the softmax denominator only;
    // a sliding window (keys [len - window, len)); and, for the extend ("prefill") form, a row of the block table per query row
    const void *sinks;          // [q_heads], element type sinks_dtype (MI_DTYPE_*); null = none
    int sinks_dtype;
    int window;                 // -1 = all keys
    const int32_t *bt_rows;     // [batch] block-table row of every query row; null = row b
    // round 4: the length-aware work list of decode_plan.h (null = uniform num_splits): unit = item, its tile range and piece index come from
    // the list, its partial lives at slot (item, head of the group)
    const int32_t *plan;
    // two-piece sequences finished by their own workgroups (gqa_decode_wide.hip): the merge kernel re-arms the meeting words, returns at once
    // when need_merge does not carry this launch's tag, and skips heads whose finishing piece marked its sum (sign bit); null = plain merge
    uint64_t *pair_flags = nullptr, *need_merge = nullptr;
    uint64_t pair_tag = 0;
};
__device__ __forceinline__ float gqa_sink_l2(const GqaParams &p, int head)      // the sink logit in the kernel's log2 domain (NOT scaled by sm_scale)
{
    float v;
    if (p.sinks_dtype == MI_DTYPE_F32) v = ((const float *)p.sinks)[head];
    else if (p.sinks_dtype == MI_DTYPE_BF16) v = __uint_as_float((uint32_t)((const uint16_t *)p.sinks)[head] << 16);
    else v = (float)__builtin_bit_cast(_Float16, ((const uint16_t *)p.sinks)[head]);
    return v * 1.4426950408889634f;
}

template <bool BF16>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}

constexpr int row_stride(int dim) { return dim * 2 + 32; }      // dim % 16 == 0 and (dim / 16) even -> odd count of 32-B units

// DKP / DVP: compile-time padded head dims (DKP % 32 == 0, DVP % 32 == 0); TILE keys per tile (64 or 32);
// HB head blocks of 16 per wave (a workgroup covers 64 * HB heads of the group).
template <bool BF16, int DKP, int DVP, int TILE, int HB>
__global__ __launch_bounds__(256) void gqa_decode_kernel(GqaParams p)
{
    constexpr int KS = row_stride(DKP), VS = row_stride(DVP);
    constexpr int kBuf = TILE * (KS + VS);
    constexpr int MT = TILE / 16, KK = TILE / 32, QS = DKP / 32, DT = DVP / 16;
    constexpr int KC = DKP / 8, VC = DVP / 8;                    // 16-B chunks per row
    constexpr int RP = TILE / 32;                                // row passes: 8 threads per row, 32 rows per pass
    constexpr int KJ = (KC + 7) / 8, VJ = (VC + 7) / 8;          // chunk passes per row
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    constexpr int kHeadsPerWg = 64 * HB;
    const int head_blocks = (p.group + kHeadsPerWg - 1) / kHeadsPerWg;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;         // sibling head blocks of a unit share an XCD (L2 reuse)
    const int unit = (jj / head_blocks) * 8 + xcd;
    const int hblk = jj % head_blocks;
    int split, nsplits, kvh, b, t_begin, t_end, seq_len, start_kv = 0;
    if (p.plan) {
        if (unit >= p.plan[0]) return;                           // (the grid is rounded up to rows of 8 workgroups)
        const mi_sgl::PlanItem it = mi_sgl::plan_item(p.plan, (long long)p.batch * p.kv_heads, unit);
        if (it.seq < 0) return;                                  // behind the list
        split = it.k, nsplits = it.n, kvh = it.seq % p.kv_heads, b = it.seq / p.kv_heads, t_begin = it.t_begin, t_end = it.t_end;
        seq_len = p.seq_lens[b];
        const int ntiles = (seq_len + TILE - 1) / TILE;          // clamped to the sequence as it is now; the last piece runs to its end
        t_begin = min(t_begin, ntiles);
        t_end = split == nsplits - 1 ? ntiles : min(t_end, ntiles);
    } else {
        if (unit >= p.batch * p.kv_heads * p.num_splits) return;
        split = unit % p.num_splits, nsplits = p.num_splits;
        kvh = (unit / p.num_splits) % p.kv_heads;
        b = unit / (p.num_splits * p.kv_heads);
        seq_len = p.seq_lens[b];
        const int ntiles = (seq_len + TILE - 1) / TILE;
        // sliding window: keys [start_kv, seq_len) (sinks_attention.py:35-39); the splits share the tiles from the window's first one on
        start_kv = (p.window >= 0 && seq_len > p.window) ? seq_len - p.window : 0;
        const int first_tile = start_kv / TILE;
        const int tps = (ntiles - first_tile + p.num_splits - 1) / p.num_splits;
        t_begin = first_tile + split * tps;
        t_end = min(ntiles, t_begin + tps);
    }
    const int64_t bt_row = p.bt_rows ? p.bt_rows[b] : b;

    // Q^T fragments: lane (g, c16) holds q[head][ks*32 + g*8 .. +8] of head block hb
    s16x8 qf[HB][QS];
    int hg[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        hg[hb] = hblk * kHeadsPerWg + (hb * 4 + wave) * 16 + c16;
        const bool ok = hg[hb] < p.group;
        const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)(kvh * p.group + (ok ? hg[hb] : 0)) * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < QS; ++ks) {
            const int d = ks * 32 + g * 8;
            qf[hb][ks] = (ok && d < p.lk) ? *(const s16x8 *)(qrow + d) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    // does this wave have any real head?  (waves without one still help to stage tiles)
    bool wave_has[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) wave_has[hb] = hblk * kHeadsPerWg + (hb * 4 + wave) * 16 < p.group;

    f32x4 acc[HB][DT];
    float m_run[HB], l_run[HB];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        m_run[hb] = -INFINITY, l_run[hb] = 0.f;
#pragma unroll
        for (int i = 0; i < DT; ++i) acc[hb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- tile staging: thread owns rows (tid >> 3) [+32] and chunks (tid & 7) + 8 j
    u32x4 kreg[RP][KJ], vreg[RP][VJ];
    const int srow = tid >> 3, sch = tid & 7;
    // block ids of the tile that stage_load fetches next: requested one tile ahead (bt_load), so that the row loads of a tile do not sit
    // behind a block-table round trip -- and never behind each other: left in one loop the compiler emitted `block id -> vmcnt(0) ->
    // rows of group 0 -> block id -> vmcnt(0) -> rows of group 1`, four serial memory round trips per tile
    const int page_shift = (p.page_size & (p.page_size - 1)) == 0 ? __builtin_ctz(p.page_size) : -1;
    auto key_of = [&](int tile, int rp) {
        int n = tile * TILE + rp * 32 + srow;
        n = n < seq_len ? n : seq_len - 1;                        // rows past the end are masked later; keep the address valid
        return n < 0 ? 0 : n;
    };
    auto page_of = [&](int n) { return page_shift >= 0 ? n >> page_shift : n / p.page_size; };
    int32_t nblk[RP];
    auto bt_load = [&](int tile) {
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) nblk[rp] = p.block_table[bt_row * p.bt_stride + page_of(key_of(tile, rp))];
    };
    auto stage_load = [&](int tile) {                             // nblk holds this tile's block ids
        const uint16_t *kr[RP], *vr[RP];
        // both ids are consumed HERE (requested a tile ago: no wait in practice); otherwise the wait for the second one lands behind the
        // first group's row loads as a vmcnt(0)
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) asm volatile("" : "+v"(nblk[rp]));
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            const int n = key_of(tile, rp);
            const int64_t blk = nblk[rp], r = n - page_of(n) * p.page_size;
            kr[rp] = p.k + blk * p.k_sblk + r * p.k_srow + (int64_t)kvh * p.k_sh;
            vr[rp] = p.v + blk * p.v_sblk + r * p.v_srow + (int64_t)kvh * p.v_sh;
        }
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int c = sch + 8 * j;
                kreg[rp][j] = (c * 8 < p.lk) ? *(const u32x4 *)(kr[rp] + c * 8) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < VJ; ++j) {
                const int c = sch + 8 * j;
                vreg[rp][j] = (c * 8 < p.lv) ? *(const u32x4 *)(vr[rp] + c * 8) : u32x4{0, 0, 0, 0};
            }
        }
        bt_load(tile + 1);                                        // clamped into the sequence: always a valid entry
    };
    auto stage_store = [&](uint8_t *buf) {
#pragma unroll
        for (int rp = 0; rp < RP; ++rp) {
            const int row = rp * 32 + srow;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int c = sch + 8 * j;
                if (c < KC) *(u32x4 *)(buf + row * KS + c * 16) = kreg[rp][j];
            }
#pragma unroll
            for (int j = 0; j < VJ; ++j) {
                const int c = sch + 8 * j;
                if (c < VC) *(u32x4 *)(buf + TILE * KS + row * VS + c * 16) = vreg[rp][j];
            }
        }
    };

    if (t_begin < t_end) {
        bt_load(t_begin);
        stage_load(t_begin);
        stage_store(lds);
    }
    // nothing the compiler knows of is in flight when the tile loop starts (Q^T above all): otherwise its wait for those registers lands
    // inside the loop as a vmcnt(0) in front of every tile's first MFMA, i.e. behind the NEXT tile's row loads
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const float cs = p.sm_scale * 1.4426950408889634f;
    for (int t = t_begin; t < t_end; ++t) {
        uint8_t *buf = lds + ((t - t_begin) & 1) * kBuf;
        uint8_t *nbuf = lds + ((t + 1 - t_begin) & 1) * kBuf;
        const bool more = t + 1 < t_end;
        __syncthreads();                                          // tile t complete in LDS; tile t-1 no longer read
        if (more) stage_load(t + 1);                              // travels under this tile's MFMAs

#pragma unroll
        for (int hb = 0; hb < HB; ++hb) {
            if (!wave_has[hb]) continue;                          // wave-uniform
            // ---- S^T[key, head] = K . Q^T
            f32x4 s[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < QS; ++ks)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const s16x8 a = *(const s16x8 *)(buf + (mt * 16 + c16) * KS + ks * 64 + g * 16);
                    s[mt] = mfma16<BF16>(a, qf[hb][ks], s[mt]);
                }
            // ---- online softmax in the scaled log2 domain; lane owns head c16 and keys mt*16 + 4g + r
            if ((t + 1) * TILE > seq_len || t * TILE < start_kv) {      // the tile that crosses the end, the tile the window starts in
                const int kbase = t * TILE + 4 * g;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kbase + mt * 16 + r >= seq_len || kbase + mt * 16 + r < start_kv) s[mt][r] = -INFINITY;
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) tmax = fmaxf(fmaxf(tmax, fmaxf(s[mt][0], s[mt][1])), fmaxf(s[mt][2], s[mt][3]));
            tmax = max_over_rows(tmax);
            tmax *= cs;                                           // sm_scale > 0: max commutes with the scaling
            if (__any(tmax > m_run[hb])) {
                const float m_new = fmaxf(m_run[hb], tmax);
                const float alpha = (m_new == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m_run[hb] - m_new);
                l_run[hb] *= alpha;
                m_run[hb] = m_new;
#pragma unroll
                for (int i = 0; i < DT; ++i) acc[hb][i] *= alpha;
            }
            const float nm = (m_run[hb] == -INFINITY) ? 0.f : -m_run[hb];
            float psum = 0.f;
            uint32_t pk[MT * 2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
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
            l_run[hb] += psum;
            // ---- O^T[d, head] += V^T . P^T ; k-step kk covers key tiles (2kk, 2kk+1): slots 0..3 / 4..7 of lane group g
            const uint8_t *vrow = buf + TILE * KS + (4 * g + (c16 >> 2)) * VS + (c16 & 3) * 8;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const s16x8 pf = __builtin_bit_cast(s16x8, u32x4{pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]});
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3))) *)(vrow + (2 * kk) * 16 * VS + dt * 32));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4 __attribute__((address_space(3))) *)(vrow + (2 * kk + 1) * 16 * VS + dt * 32));
                    const s16x8 a = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    acc[hb][dt] = mfma16<BF16>(a, pf, acc[hb][dt]);
                }
            }
        }
        if (more) stage_store(nbuf);
    }

    // ---- epilogue: lane holds O^T[d = dt*16 + 4g + r][head c16]
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
        if (hg[hb] >= p.group) continue;
        const int head = kvh * p.group + hg[hb];
        if (nsplits == 1) {
            float inv;
            if (p.sinks) {                                        // l += exp(sink - max) with the sink inside the max (:78-80)
                const float sk = gqa_sink_l2(p, head);
                const float M = fmaxf(m_run[hb], sk);
                const float wa = (m_run[hb] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run[hb] - M);
                const float L = l_run[hb] * wa + __builtin_amdgcn_exp2f(sk - M);
                inv = wa / L;
            } else {
                inv = l_run[hb] > 0.f ? 1.f / l_run[hb] : 0.f;
            }
            uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)head * p.o_sh + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (dt * 16 + 4 * g >= p.lv) continue;            // lv % 8 == 0: the 4 dims are in or out together
                const uint32_t w0 = pack2<BF16>(acc[hb][dt][0] * inv, acc[hb][dt][1] * inv);
                const uint32_t w1 = pack2<BF16>(acc[hb][dt][2] * inv, acc[hb][dt][3] * inv);
                *(uint2 *)(orow + dt * 16) = uint2{w0, w1};
            }
        } else {
            const int64_t idx = p.plan ? (int64_t)unit * p.group + hg[hb] : ((int64_t)b * p.q_heads + head) * p.num_splits + split;
            float *po = p.ws_o + idx * DVP + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) *(f32x4 *)(po + dt * 16) = acc[hb][dt];
            if (g == 0) {
                p.ws_ml[idx * 2 + 0] = m_run[hb];
                p.ws_ml[idx * 2 + 1] = l_run[hb];
            }
        }
    }
}

// merge the flash-decoding partials: one wave per (b, head); lane handles dims lane*4 + 256 i
template <bool BF16>
__global__ __launch_bounds__(256) void gqa_merge_kernel(GqaParams p, int dvp)
{
    const int lane = threadIdx.x & 63;
    const int64_t bh = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (bh >= (int64_t)p.batch * p.q_heads) return;
    const int b = (int)(bh / p.q_heads), h = (int)(bh % p.q_heads);
    // the meeting words of the sequence's two pieces (gqa_decode_wide.hip) are re-armed here, behind the launch that used them: the next
    // call -- or the next replay of a captured one, which carries the same tag -- finds them clear
    if (p.pair_flags && h % p.group == 0 && lane < 2) p.pair_flags[((int64_t)b * p.kv_heads + h / p.group) * 2 + lane] = 0ull;
    // every sequence finished in the decode kernel (two pieces each, or one): one load per wave instead of the statistics round trips
    if (p.need_merge && __hip_atomic_load(p.need_merge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.pair_tag) return;
    // partial s of this head: slot (bh, s), or -- planned form -- slot (item of piece s, head within the group)
    int S = p.num_splits, first = 0;
    if (p.plan) {
        const int32_t *info = p.plan + mi_sgl::kPlanHdr + 2ll * (b * p.kv_heads + h / p.group);
        first = info[0], S = info[1];
        if (S == 1) return;                                   // the piece wrote the output row itself
    }
    auto slot = [&](int s) -> int64_t { return p.plan ? (int64_t)(first + s) * p.group + h % p.group : bh * S + s; };
    // two pieces: piece (head within the group) / 64 finished this head itself if its sum carries the mark (the sign); without the mark
    // (its partner did not show up in time) both partials of the head are complete in the workspace: merge as usual
    const bool marks = p.pair_flags != nullptr && S == 2;
    if (marks && (__float_as_uint(p.ws_ml[slot((h % p.group) >> 6) * 2 + 1]) >> 31)) return;
    auto sum_of = [&](int s) -> float { const float l = p.ws_ml[slot(s) * 2 + 1]; return marks ? fabsf(l) : l; };
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, p.ws_ml[slot(s) * 2]);
    const float sk = p.sinks ? gqa_sink_l2(p, h) : -INFINITY;
    M = fmaxf(M, sk);
    float L = p.sinks ? __builtin_amdgcn_exp2f(sk - M) : 0.f;
    // (explicit fused multiply-adds: gqa_decode_wide.hip's pair finish forms the same sums in the same order and must round the same way)
    for (int s = 0; s < S; ++s) {
        const float m = p.ws_ml[slot(s) * 2];
        if (m != -INFINITY) L = __builtin_fmaf(__builtin_amdgcn_exp2f(m - M), sum_of(s), L);
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh;
    for (int d = lane * 4; d < p.lv; d += 256) {
        f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            const float m = p.ws_ml[slot(s) * 2];
            if (m == -INFINITY) continue;
            const float w = __builtin_amdgcn_exp2f(m - M);
            const f32x4 a = *(const f32x4 *)(p.ws_o + slot(s) * dvp + d);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = __builtin_fmaf(w, a[i], o[i]);
        }
        *(uint2 *)(orow + d) = uint2{pack2<BF16>(o[0] * inv, o[1] * inv), pack2<BF16>(o[2] * inv, o[3] * inv)};
    }
}

struct Shape {
    int dkp, dvp, tile;
};
// first entry that fits is used; (576, 512) covers DeepSeek-style K = nope|rope caches whose V is NOT a view of K
static const Shape kShapes[] = {{64, 64, 64}, {128, 128, 64}, {192, 128, 64}, {256, 256, 64}, {288, 256, 64}, {576, 512, 32}};

static const Shape *pick_shape(int lk, int lv)
{
    for (const Shape &s : kShapes)
        if (lk <= s.dkp && lv <= s.dvp) return &s;
    return nullptr;
}

template <bool BF16, int DKP, int DVP, int TILE, int HB>
static void launch_one(const GqaParams &p, dim3 grid, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)TILE * (row_stride(DKP) + row_stride(DVP));
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute((const void *)gqa_decode_kernel<BF16, DKP, DVP, TILE, HB>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    gqa_decode_kernel<BF16, DKP, DVP, TILE, HB><<<grid, 256, lds, st>>>(p);
}

template <bool BF16, int HB>
static bool launch_shape(const Shape &s, const GqaParams &p, dim3 grid, hipStream_t st)
{
    if (s.dkp == 64) launch_one<BF16, 64, 64, 64, HB>(p, grid, st);
    else if (s.dkp == 128) launch_one<BF16, 128, 128, 64, HB>(p, grid, st);
    else if (s.dkp == 192) launch_one<BF16, 192, 128, 64, HB>(p, grid, st);
    else if (s.dkp == 256) launch_one<BF16, 256, 256, 64, HB>(p, grid, st);
    else if (s.dkp == 288) launch_one<BF16, 288, 256, 64, HB>(p, grid, st);
    else if (s.dkp == 576) launch_one<BF16, 576, 512, 32, 1>(p, grid, st);
    else return false;
    return true;
}

static int heads_per_wg(const Shape &s, int group) { return (s.dkp == 576 || group <= 64) ? 64 : 128; }

}  // namespace mi_gqa

using namespace mi_gqa;

// ---- the planned form (decode_plan.h): a device-built, length-aware work list instead of one split count for every sequence.  Served for
// plain decode (no sinks, no sliding window, no per-row block tables); MI_GQA_PLAN=0 keeps uniform splits.
static int gqa_cus()
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
constexpr int kGqaMaxWgPerCu = 4;
// rows of partials a planned workspace holds: (items) x (heads of a group) <= batch * q_heads + (workers + padding) * q_heads
static size_t gqa_plan_rows_cap(int batch, int q_heads) { return (size_t)batch * q_heads + (size_t)(gqa_cus() * kGqaMaxWgPerCu + 8) * q_heads; }
static size_t gqa_plan_words_cap(int batch, int q_heads) { return mi_sgl::plan_words((long long)batch * q_heads, gqa_cus() * kGqaMaxWgPerCu); }
static bool gqa_plan_allowed()
{
    static const bool allow = !(getenv("MI_GQA_PLAN") && atoi(getenv("MI_GQA_PLAN")) == 0);
    return allow;
}

extern "C" size_t mi_gqa_decode_workspace(int batch, int q_heads, int v_dim, int num_splits)
{
    if (num_splits == MI_MLA_SPLITS_PLANNED) {
        if (batch <= 0 || q_heads <= 0) return 0;
        return gqa_plan_rows_cap(batch, q_heads) * (512 + 2) * sizeof(float) + gqa_plan_words_cap(batch, q_heads) * sizeof(int32_t);
    }
    if (num_splits <= 1) return 0;
    const Shape *s = pick_shape(8, v_dim);
    if (!s) return 0;
    // the padded V width depends on the (k_dim, v_dim) pair; 512 bounds every supported shape
    return (size_t)batch * q_heads * num_splits * (512 + 2) * sizeof(float);
}

extern "C" int mi_gqa_decode_num_splits(int batch, int q_heads, int kv_heads, int max_seq_len)
{
    if (batch <= 0 || q_heads <= 0 || kv_heads <= 0 || max_seq_len <= 0) return 1;
    const long long wgs = (long long)batch * kv_heads * ((q_heads / kv_heads + 127) / 128);
    // (The planned form -- num_splits = MI_MLA_SPLITS_PLANNED, decode_plan.h -- is served but never chosen here.  Measured on Llama-70B-shaped
    //  decode (64 / 8 heads, d = 128; tools/probes/gqa_plan_ab.py): 16 sequences x 8192 keys 113.9 us planned vs 105.0 us with four uniform
    //  splits (ragged 65.1 vs 64.2); 64 x 4096 208 vs 190 us (ragged 144 vs 118-130).  This kernel's workgroups are light and two to three
    //  share a CU, so uniform splits already balance; "all pieces in one round" -- right for the MLA kernels, one heavy workgroup per CU --
    //  leaves 512 sequences on 512 slots unsplit and pays a plan launch and a merge launch for nothing.)
    const int ntiles = (max_seq_len + 63) / 64;
    int s = (int)((512 + wgs - 1) / wgs);              // about two workgroups per CU
    const int cap = ntiles / 4 > 1 ? ntiles / 4 : 1;   // keep >= 4 tiles (256 keys) per split
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

static int g_gqa_pair_mode = -1;    // -1 = environment / default (on); 0 = off, 1 = on, 2 = on with piece 1 withholding its word (tests)
extern "C" int mi_gqa_decode_set_pair(int mode)
{
    if (mode < -1 || mode > 2) return MI_SGL_EINVAL;
    g_gqa_pair_mode = mode;
    return MI_SGL_OK;
}

static int gqa_decode_impl(const void *q, const void *k, const void *v, void *out, const int32_t *kv_seq_lens,
                             const int32_t *block_table, int batch, int q_heads, int kv_heads, int k_dim, int v_dim, int page_size,
                             int bt_stride, int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_blk,
                             int64_t k_stride_row, int64_t k_stride_h, int64_t v_stride_blk, int64_t v_stride_row,
                             int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, int num_splits,
                             void *workspace, size_t workspace_bytes, void *stream, const void *sinks, int sinks_dtype, int window,
                             const int32_t *bt_rows)
{
    if (batch < 0 || q_heads <= 0 || kv_heads <= 0 || q_heads % kv_heads || page_size <= 0 || bt_stride <= 0) return MI_SGL_EINVAL;
    if (k_dim <= 0 || v_dim <= 0 || (k_dim % 8) || (v_dim % 8)) return MI_SGL_EINVAL;
    const Shape *shape = pick_shape(k_dim, v_dim);
    if (!shape) return MI_SGL_EINVAL;
    if (batch == 0) return MI_SGL_OK;
    if (!q || !k || !v || !out || !kv_seq_lens || !block_table) return MI_SGL_EINVAL;
    if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_SGL_EINVAL;
    if ((q_stride_h % 8) || (q_stride_b % 8) || (k_stride_row % 8) || (k_stride_blk % 8) || (k_stride_h % 8) || (v_stride_row % 8) ||
        (v_stride_blk % 8) || (v_stride_h % 8) || (o_stride_h % 4) || (o_stride_b % 4))
        return MI_SGL_EINVAL;      // 16-byte loads, 8-byte stores
    if (num_splits == 0) num_splits = mi_gqa_decode_num_splits(batch, q_heads, kv_heads, max_seq_len);
    if (num_splits < 0 && num_splits != MI_MLA_SPLITS_PLANNED) return MI_SGL_EINVAL;
    bool planned = num_splits == MI_MLA_SPLITS_PLANNED;
    if (planned && (sinks || window >= 0 || bt_rows || !gqa_plan_allowed())) {
        // the planned form does not serve this call: uniform splits, as many as the caller's workspace holds
        planned = false;
        const long long wgs = (long long)batch * kv_heads * ((q_heads / kv_heads + 127) / 128);
        const int ntiles = (max_seq_len + 63) / 64, cap = ntiles / 4 > 1 ? ntiles / 4 : 1;
        num_splits = (int)std::min<long long>(std::min<long long>((512 + wgs - 1) / wgs, cap), 64);
        if (num_splits < 1) num_splits = 1;
        while (num_splits > 1 && mi_gqa_decode_workspace(batch, q_heads, v_dim, num_splits) > workspace_bytes) --num_splits;
    }
    if ((num_splits > 1 || planned) &&
        (!workspace || workspace_bytes < mi_gqa_decode_workspace(batch, q_heads, v_dim, planned ? MI_MLA_SPLITS_PLANNED : num_splits)))
        return MI_SGL_EINVAL;
    // Large kv groups (more than 64 query heads per kv head) with head dims up to (288, 256): the eight-wave LDS-DMA kernel of
    // gqa_decode_wide.hip -- one heavy workgroup per CU, so the split count is the MLA kernels' (fill the CUs once), never more than asked for.
    if (!sinks && window < 0 && !bt_rows &&
        mi_gqa_wide::applies(q_heads / kv_heads, k_dim, v_dim, page_size, k_stride_blk, k_stride_row, v_stride_blk, v_stride_row)) {
        mi_gqa_wide::Params w;
        w.q = (const uint16_t *)q, w.k = (const uint16_t *)k, w.v = (const uint16_t *)v, w.out = (uint16_t *)out;
        w.seq_lens = kv_seq_lens, w.block_table = block_table;
        w.batch = batch, w.q_heads = q_heads, w.kv_heads = kv_heads, w.group = q_heads / kv_heads, w.page_size = page_size;
        w.bt_stride = bt_stride, w.lk = k_dim, w.lv = v_dim;
        w.q_sb = q_stride_b, w.q_sh = q_stride_h, w.k_sblk = k_stride_blk, w.k_srow = k_stride_row, w.k_sh = k_stride_h;
        w.v_sblk = v_stride_blk, w.v_srow = v_stride_row, w.v_sh = v_stride_h, w.o_sb = o_stride_b, w.o_sh = o_stride_h;
        w.sm_scale = sm_scale, w.plan = nullptr;
        const int head_blocks = (w.group + 127) / 128;
        hipStream_t st = (hipStream_t)stream;
        if (!planned) {
            const long long wgs = (long long)batch * kv_heads * head_blocks;
            const int ntiles = (max_seq_len + mi_gqa_wide::kTile - 1) / mi_gqa_wide::kTile, cap = ntiles / 8 > 1 ? ntiles / 8 : 1;   // >= 256 keys per split
            const int fill = (int)std::min<long long>(std::min<long long>((gqa_cus() + wgs - 1) / wgs, cap), 64);
            num_splits = std::max(1, std::min(num_splits, fill));
        }
        w.num_splits = num_splits;
        w.ws_o = (float *)workspace;
        w.ws_ml = w.ws_o ? w.ws_o + (planned ? gqa_plan_rows_cap(batch, q_heads) : (size_t)batch * q_heads * num_splits) * mi_gqa_wide::kDVP : nullptr;
        long long units = (long long)batch * kv_heads * num_splits;
        if (planned) {
            const int workers = std::max(1, gqa_cus() / head_blocks);
            int32_t *plan = (int32_t *)(w.ws_ml + gqa_plan_rows_cap(batch, q_heads) * 2);
            mi_sgl::decode_plan_kernel<<<1, 1024, 0, st>>>(kv_seq_lens, batch, kv_heads, mi_gqa_wide::tile_keys(w), 1, workers, plan);
            w.plan = plan;
            w.num_splits = num_splits = 1;
            units = mi_sgl::plan_items_max((long long)batch * kv_heads, workers);
        }
        // Two-piece sequences finish between their two workgroups (gqa_wide.h): the meeting words live behind the partial area / the work list,
        // inside what mi_gqa_decode_workspace() sizes for 512-float rows (this kernel's are 256); tagged per call, re-armed by the merge kernel
        w.pair_flags = nullptr, w.need_merge = nullptr, w.pair_tag = 0, w.pair_withhold = 0;
        static const bool pair_env = !(getenv("MI_GQA_PAIR") && atoi(getenv("MI_GQA_PAIR")) == 0);
        const bool pair_on = g_gqa_pair_mode < 0 ? pair_env : g_gqa_pair_mode != 0;
        if (pair_on && head_blocks == 1 && (planned || num_splits == 2)) {
            const char *area = planned ? (const char *)(w.plan + gqa_plan_words_cap(batch, q_heads))
                                       : (const char *)(w.ws_ml + (size_t)batch * q_heads * num_splits * 2);
            uint64_t *flags = (uint64_t *)(((uintptr_t)area + 7) & ~(uintptr_t)7);
            const size_t words = 2 * (size_t)batch * kv_heads + 1;
            if ((const char *)(flags + words) <= (const char *)workspace + workspace_bytes) {
                static uint32_t epoch = 0;
                const uint32_t e = ++epoch ? epoch : ++epoch;
                w.pair_flags = flags, w.need_merge = flags + (words - 1);
                w.pair_tag = ((uint64_t)e * 0x9E3779B97F4A7C15ull) | 1ull;      // never 0
                w.pair_withhold = g_gqa_pair_mode == 2;
            }
        }
        mi_gqa_wide::launch(w, dtype, units, st);
        if (num_splits > 1 || planned) {
            GqaParams m{};                                       // the merge kernel reads the partial layout the wide kernel wrote (row = kDVP floats)
            m.pair_flags = w.pair_flags, m.need_merge = w.need_merge, m.pair_tag = w.pair_tag;
            m.out = (uint16_t *)out, m.ws_o = w.ws_o, m.ws_ml = w.ws_ml, m.batch = batch, m.q_heads = q_heads, m.kv_heads = kv_heads;
            m.group = w.group, m.num_splits = num_splits, m.lv = v_dim, m.o_sb = o_stride_b, m.o_sh = o_stride_h, m.plan = w.plan;
            m.sinks = nullptr, m.window = -1, m.bt_rows = nullptr;
            const long long bh = (long long)batch * q_heads;
            const int blocks = (int)((bh + 3) / 4);
            if (dtype == MI_DTYPE_BF16) gqa_merge_kernel<true><<<blocks, 256, 0, st>>>(m, mi_gqa_wide::kDVP);
            else gqa_merge_kernel<false><<<blocks, 256, 0, st>>>(m, mi_gqa_wide::kDVP);
        }
        return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
    }
    GqaParams p;
    p.plan = nullptr;
    p.q = (const uint16_t *)q, p.k = (const uint16_t *)k, p.v = (const uint16_t *)v;
    p.out = (uint16_t *)out, p.seq_lens = kv_seq_lens, p.block_table = block_table;
    p.ws_o = (float *)workspace;
    p.ws_ml = p.ws_o ? p.ws_o + (planned ? gqa_plan_rows_cap(batch, q_heads) : (size_t)batch * q_heads * num_splits) * shape->dvp : nullptr;
    p.batch = batch, p.q_heads = q_heads, p.kv_heads = kv_heads, p.group = q_heads / kv_heads, p.page_size = page_size;
    p.bt_stride = bt_stride, p.num_splits = num_splits, p.lk = k_dim, p.lv = v_dim;
    p.q_sb = q_stride_b, p.q_sh = q_stride_h, p.k_sblk = k_stride_blk, p.k_srow = k_stride_row, p.k_sh = k_stride_h;
    p.v_sblk = v_stride_blk, p.v_srow = v_stride_row, p.v_sh = v_stride_h, p.o_sb = o_stride_b, p.o_sh = o_stride_h;
    p.sm_scale = sm_scale;
    p.sinks = sinks, p.sinks_dtype = sinks_dtype, p.window = window, p.bt_rows = bt_rows;
    hipStream_t st = (hipStream_t)stream;
    const int hpw = heads_per_wg(*shape, p.group);
    const int head_blocks = (p.group + hpw - 1) / hpw;
    long long units = (long long)batch * kv_heads * num_splits;
    if (planned) {
        // workgroups the chip runs at once: as many per CU as the LDS of this shape allows; a (sequence, kv head) piece is head_blocks of them
        const size_t lds = 2 * (size_t)shape->tile * ((size_t)shape->dkp * 2 + 32 + (size_t)shape->dvp * 2 + 32);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(kGqaMaxWgPerCu, (160 * 1024) / lds));
        const int workers = std::max(1, gqa_cus() * per_cu / head_blocks);
        int32_t *plan = (int32_t *)(p.ws_ml + gqa_plan_rows_cap(batch, q_heads) * 2);
        mi_sgl::decode_plan_kernel<<<1, 1024, 0, st>>>(kv_seq_lens, batch, kv_heads, shape->tile, 1, workers, plan);
        p.plan = plan;
        p.num_splits = num_splits = 1;
        units = mi_sgl::plan_items_max((long long)batch * kv_heads, workers);
    }
    dim3 grid((unsigned)(((units + 7) / 8) * 8 * head_blocks));
    bool ok;
    if (dtype == MI_DTYPE_BF16) ok = hpw == 64 ? launch_shape<true, 1>(*shape, p, grid, st) : launch_shape<true, 2>(*shape, p, grid, st);
    else ok = hpw == 64 ? launch_shape<false, 1>(*shape, p, grid, st) : launch_shape<false, 2>(*shape, p, grid, st);
    if (!ok) return MI_SGL_EINVAL;
    if (num_splits > 1 || planned) {
        const long long bh = (long long)batch * q_heads;
        const int blocks = (int)((bh + 3) / 4);
        if (dtype == MI_DTYPE_BF16) gqa_merge_kernel<true><<<blocks, 256, 0, st>>>(p, shape->dvp);
        else gqa_merge_kernel<false><<<blocks, 256, 0, st>>>(p, shape->dvp);
    }
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_gqa_decode(const void *q, const void *k, const void *v, void *out, const int32_t *kv_seq_lens,
                             const int32_t *block_table, int batch, int q_heads, int kv_heads, int k_dim, int v_dim, int page_size,
                             int bt_stride, int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_blk,
                             int64_t k_stride_row, int64_t k_stride_h, int64_t v_stride_blk, int64_t v_stride_row,
                             int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, int num_splits,
                             void *workspace, size_t workspace_bytes, void *stream)
{
    return gqa_decode_impl(q, k, v, out, kv_seq_lens, block_table, batch, q_heads, kv_heads, k_dim, v_dim, page_size, bt_stride, max_seq_len,
                           q_stride_b, q_stride_h, k_stride_blk, k_stride_row, k_stride_h, v_stride_blk, v_stride_row, v_stride_h, o_stride_b,
                           o_stride_h, sm_scale, dtype, num_splits, workspace, workspace_bytes, stream, nullptr, 0, -1, nullptr);
}

extern "C" int mi_gqa_decode_sinks(const void *q, const void *k, const void *v, void *out, const int32_t *kv_seq_lens,
                                   const int32_t *block_table, int batch, int q_heads, int kv_heads, int k_dim, int v_dim, int page_size,
                                   int bt_stride, int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_blk,
                                   int64_t k_stride_row, int64_t k_stride_h, int64_t v_stride_blk, int64_t v_stride_row,
                                   int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, int num_splits,
                                   void *workspace, size_t workspace_bytes, const void *sinks, int sinks_dtype, int sliding_window,
                                   const int32_t *block_table_rows, void *stream)
{
    if (sinks && sinks_dtype != MI_DTYPE_BF16 && sinks_dtype != MI_DTYPE_F16 && sinks_dtype != MI_DTYPE_F32) return MI_SGL_EINVAL;
    if (sliding_window < -1) return MI_SGL_EINVAL;
    return gqa_decode_impl(q, k, v, out, kv_seq_lens, block_table, batch, q_heads, kv_heads, k_dim, v_dim, page_size, bt_stride, max_seq_len,
                           q_stride_b, q_stride_h, k_stride_blk, k_stride_row, k_stride_h, v_stride_blk, v_stride_row, v_stride_h, o_stride_b,
                           o_stride_h, sm_scale, dtype, num_splits, workspace, workspace_bytes, stream, sinks, sinks_dtype, sliding_window,
                           block_table_rows);
}


// ---- per-query block tables of the sparse + causal prefill path (attention/fia_blockq_attention.py:11-88) ------------------------------------
// One wave per query (topk + 1 <= 64 slots, a lane each): the query's own logical block (the one holding its position) goes LAST, the other
// real blocks keep their order in front of it, every logical block becomes the physical page req_to_token[req, block * block_size] /
// block_size, pads are 0; actual_kvlen = real non-own blocks * block_size (+ offset in the own block + 1 when the own block is selected)
// (:35-88).  The attention itself is then a paged decode with one query row per "sequence" (mi_gqa_decode): the reference hands the same
// tables to its fused-infer-attention op (:167-180), after a host round trip for the lengths that this path does not need.
namespace mi_sgl {
__global__ __launch_bounds__(256) void fia_prep_kernel(const int32_t *__restrict__ topk_idx, long long stride_ti_t, long long stride_ti_k,
                                                       const int32_t *__restrict__ seq_lens, const void *__restrict__ req_pool, int req_is_i64,
                                                       const int32_t *__restrict__ req_to_token, long long stride_rtt_r, long long stride_rtt_t,
                                                       int max_cols, int total_q, int topk1, int block_size, int32_t *__restrict__ block_table,
                                                       long long stride_bt_t, int32_t *__restrict__ actual_kvlen)
{
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= total_q) return;
    const int seq_len = seq_lens[q];
    const int abs_pos = max(seq_len - 1, 0);
    const int own = abs_pos / block_size, own_offset = abs_pos % block_size;
    const long long req = req_is_i64 ? ((const long long *)req_pool)[q] : (long long)((const int *)req_pool)[q];
    const bool slot = lane < topk1;
    const int log_blk = slot ? topk_idx[q * stride_ti_t + lane * stride_ti_k] : -1;
    const bool is_own = slot && log_blk >= 0 && log_blk == own;
    const bool real_nonown = slot && log_blk >= 0 && log_blk != own;
    const unsigned long long m = __ballot(real_nonown);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));         // exclusive scan: this slot's place among the real non-own blocks
    const int count_real = __popcll(m);
    const bool own_present = __ballot(is_own) != 0ull;
    int phys = 0;
    if (slot && log_blk >= 0) {
        const int col = min(log_blk * block_size, max_cols - 1);
        phys = req_to_token[req * stride_rtt_r + col * stride_rtt_t] / block_size;
    }
    const int own_col = min(own * block_size, max_cols - 1);
    const int own_phys = req_to_token[req * stride_rtt_r + own_col * stride_rtt_t] / block_size;
    int32_t *bt = block_table + q * stride_bt_t;
    if (slot) bt[lane] = 0;                                           // the wave's stores to one address keep program order
    if (real_nonown) bt[rank] = phys;
    // (the reference stores the own page unconditionally, also at slot topk1 -- behind the row -- when every slot is a real non-own block)
    if (lane == 0 && count_real < topk1) bt[count_real] = own_phys;
    if (lane == 0) actual_kvlen[q] = own_present ? count_real * block_size + own_offset + 1 : count_real * block_size;
}
}  // namespace mi_sgl

extern "C" int mi_fia_prep(const int32_t *topk_idx, long long stride_ti_t, long long stride_ti_k, const int32_t *seq_lens, const void *per_query_req,
                           int req_is_i64, const int32_t *req_to_token, long long stride_rtt_r, long long stride_rtt_t, int max_cols, int total_q,
                           int topk1, int block_size, int32_t *block_table, long long stride_bt_t, int32_t *actual_kvlen, void *stream)
{
    if (total_q < 0 || topk1 <= 0 || topk1 > 64 || block_size <= 0 || max_cols <= 0) return MI_SGL_EINVAL;
    if (total_q == 0) return MI_SGL_OK;
    if (!topk_idx || !seq_lens || !per_query_req || !req_to_token || !block_table || !actual_kvlen) return MI_SGL_EINVAL;
    mi_sgl::fia_prep_kernel<<<(total_q + 3) / 4, 256, 0, (hipStream_t)stream>>>(topk_idx, stride_ti_t, stride_ti_k, seq_lens, per_query_req, req_is_i64,
                                                                                req_to_token, stride_rtt_r, stride_rtt_t, max_cols, total_q, topk1,
                                                                                block_size, block_table, stride_bt_t, actual_kvlen);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}
