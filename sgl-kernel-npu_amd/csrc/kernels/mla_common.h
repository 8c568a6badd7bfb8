// Shared pieces of the two MLA decode kernels (mla_decode.hip: 64 heads per workgroup on 16x16x32 MFMAs;
// mla_decode_wide.hip: 128 heads per workgroup on 32x32x16 MFMAs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "decode_plan.h"

namespace mi_sgl {

constexpr int kDN = 512, kDR = 64, kTile = 64;
constexpr int kNopeStride = kDN * 2 + 32;          // bytes per key row in LDS: 66 x 16-B slots, 66 mod 16 = 2 makes both the
                                                   // ds_read_b128 (16 keys x 16 B) and the tr-read (8 keys x 32 B) footprints conflict-free
constexpr int kRopeStride = kDR * 2;               // 128 B, swizzled

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct MlaParams {
    const uint16_t *q, *k_nope, *k_rope;
    uint16_t *out;
    const int32_t *seq_lens, *block_table;
    float *ws_o;      // [B][Hq][S][512] fp32 partial (unnormalised) outputs
    float *ws_ml;     // [B][Hq][S][2]   running max, running sum
    int batch, q_heads, kv_heads, group, page_size, bt_stride, num_splits;
    int64_t q_sb, q_sh, kn_sblk, kn_srow, kn_sh, kr_sblk, kr_srow, kr_sh, o_sb, o_sh;
    float sm_scale;
    // wide kernel -> merge kernel hand-off: sequences whose softmax reference was outgrown (see mla_decode_wide.hip)
    uint32_t *fix_flags;      // [batch * kv_heads], entry == fix_epoch means "recompute"
    uint32_t fix_epoch;
    int fix_only;             // merge kernel: check the hand-off words and recompute flagged sequences
    // wide kernel with in-kernel merge (num_splits <= 2): the two workgroups of a (sequence, kv head, head block) meet at
    // arrive[...]; the second one merges and, if the sequence is flagged, recomputes -- no merge launch
    uint32_t *arrive;         // [batch * kv_heads * head_blocks], values are tagged with fix_epoch (no clearing needed)
    int inline_merge;
    // eight-wave scalar-id kernel, sequences cut in TWO pieces (BASELINE C4): the two workgroups publish their partials, meet at
    // pair_flags[2 seq + piece] and each finishes one half of the output dimensions itself (mla_decode_wide8s.hip); the merge kernel
    // skips such a sequence and re-arms its two words.  NULL = every cut sequence goes through the merge kernel.
    uint64_t *pair_flags;     // [batch * kv_heads][2], a word holds pair_tag once its piece's partial is visible
    uint64_t pair_tag;        // never 0; a scrambled call number (first use of an uninitialised workspace: 2^-64 per word)
    uint64_t *need_merge;     // one word behind the pair words: holds pair_tag when some workgroup of this launch left work for the merge kernel (a
                              // sequence in three or more pieces, a pair that did not meet, an outgrown softmax reference); otherwise every
                              // merge workgroup leaves after ONE load instead of the list lookup + statistics round trips per head
    int pair_withhold;        // test hook: piece 1 never raises its word, so piece 0 runs into the bounded wait (the merge kernel's turn)
    // Length-aware work list built on the device by decode_plan_kernel (decode_plan.h), NULL = the uniform num_splits form.  Layout below.
    const int32_t *plan;
};

template <bool BF16>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    // one v_cvt_pk_{bf16,f16}_f32 (round to nearest even)
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}

template <bool BF16>
__device__ __forceinline__ uint16_t cvt_out(float f)
{
    if constexpr (BF16) {
        uint32_t x = __float_as_uint(f);
        if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
        return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16);
    } else {
        // the fp32 value is rounded to fp32 FIRST, then to fp16, wherever this is called: left alone the compiler folds a preceding
        // multiplication into v_fma_mixlo_f16 (one rounding) in some kernels and not in others (v_mul + v_cvt_pk_f16_f32), and the
        // same sums then differ by an fp16 ulp in a few elements per million between the in-kernel and the merge-kernel finish
        asm volatile("" : "+v"(f));
        _Float16 a = (_Float16)f;
        return __builtin_bit_cast(uint16_t, a);
    }
}

// Per-tile row addressing.  Lane l owns key l of a tile: tile_rows() turns its token index into byte offsets of the
// key's nope / rope rows (one block-table load per lane per tile, issued a whole tile ahead of its use so the load
// latency never sits in front of the LDS-DMA).  issue_tile() then fetches row addresses from lanes with readlane /
// shuffle, so no vector-memory wait separates consecutive DMA instructions.
struct TileRows {
    int64_t nope, rope;     // element offsets into k_nope / k_rope
};
struct TileRowsRaw {        // result of the block-table load, not yet consumed (so no wait is placed at the load)
    int blk, row;
};

__device__ __forceinline__ TileRowsRaw tile_rows_load(const MlaParams &p, int b, int seq_len, int tile, int lane)
{
    int n = tile * kTile + lane;
    n = n < seq_len ? n : seq_len - 1;                 // rows past the end are masked later; keep the address valid
    n = n < 0 ? 0 : n;
    const int page = n / p.page_size;
    TileRowsRaw r;
    r.row = n - page * p.page_size;
    r.blk = p.block_table[(int64_t)b * p.bt_stride + page];
    return r;
}

__device__ __forceinline__ TileRows tile_rows_finish(const MlaParams &p, int kvh, const TileRowsRaw &raw)
{
    TileRows r;
    r.nope = (int64_t)raw.blk * p.kn_sblk + (int64_t)raw.row * p.kn_srow + (int64_t)kvh * p.kn_sh;
    r.rope = (int64_t)raw.blk * p.kr_sblk + (int64_t)raw.row * p.kr_srow + (int64_t)kvh * p.kr_sh;
    return r;
}

__device__ __forceinline__ int64_t lane_i64(int64_t v, int src_lane)
{
    const int lo = __shfl((int)(v & 0xFFFFFFFFll), src_lane, 64), hi = __shfl((int)(v >> 32), src_lane, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}


// wide kernel (mla_decode_wide.hip): launch for kv groups of more than 64 heads
constexpr int kWideTile = 32;
void launch_mla_wide(const MlaParams &p, int dtype, long long units, hipStream_t st);
// eight-wave form (mla_decode_wide8.hip): two waves per SIMD; the default for kv groups of more than 64 heads
void launch_mla_wide8(const MlaParams &p, int dtype, long long units, hipStream_t st);
// the same work split with one scalar block id per tile and three tiles in flight (mla_decode_wide8s.hip); power-of-two pages of >= 32 keys
void launch_mla_wide8s(const MlaParams &p, int dtype, long long units, hipStream_t st);

// Slow path behind the wide kernel (mla_decode_wide.hip): a sequence whose scores outgrew the fixed softmax reference is
// recomputed here, one wave per (b, head), with plain loads and fp32 VALU math -- exact two-pass softmax (max first), P
// rounded to the KV dtype before P.V like the MFMA kernels.  Lane l owns output dims 8 l .. 8 l + 7.  Rare by construction
// (bf16: a later tile must beat the first by 2^64), so it is written for clarity, not speed.
template <bool BF16>
__device__ __forceinline__ float ld_elem(const uint16_t *ptr)
{
    if constexpr (BF16) return __uint_as_float((uint32_t)*ptr << 16);
    else return (float)__builtin_bit_cast(_Float16, *ptr);
}

template <bool BF16>
__device__ inline void mla_recompute_head(const MlaParams &p, int b, int h, int lane)
{
    const int kvh = h / p.group, seq_len = p.seq_lens[b];
    const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh;
    float qv[9];                                               // lane holds q dims lane + 64 j
#pragma unroll
    for (int j = 0; j < 9; ++j) qv[j] = ld_elem<BF16>(qrow + lane + 64 * j);
    auto key_ptrs = [&](int n, const uint16_t *&kn, const uint16_t *&kr) {
        const int page = n / p.page_size, row = n - page * p.page_size;
        const int64_t blk = p.block_table[(int64_t)b * p.bt_stride + page];
        kn = p.k_nope + blk * p.kn_sblk + (int64_t)row * p.kn_srow + (int64_t)kvh * p.kn_sh;
        kr = p.k_rope + blk * p.kr_sblk + (int64_t)row * p.kr_srow + (int64_t)kvh * p.kr_sh;
    };
    auto score = [&](int n) -> float {
        const uint16_t *kn, *kr;
        key_ptrs(n, kn, kr);
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) d += qv[j] * ld_elem<BF16>(kn + lane + 64 * j);
        d += qv[8] * ld_elem<BF16>(kr + lane);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        return d * p.sm_scale;
    };
    float m = -INFINITY;
    for (int n = 0; n < seq_len; ++n) m = fmaxf(m, score(n));
    float l = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int n = 0; n < seq_len; ++n) {
        const float pr = __expf(score(n) - m);
        l += pr;
        float prq;                                             // P in the KV dtype
        if constexpr (BF16) prq = __uint_as_float((uint32_t)cvt_out<true>(pr) << 16);
        else prq = (float)(_Float16)pr;
        const uint16_t *kn, *kr;
        key_ptrs(n, kn, kr);
        const u32x4 v = *(const u32x4 *)(kn + lane * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint16_t lo = (uint16_t)(v[j] & 0xFFFFu), hi = (uint16_t)(v[j] >> 16);
            o[2 * j] += prq * ld_elem<BF16>(&lo);
            o[2 * j + 1] += prq * ld_elem<BF16>(&hi);
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh + lane * 8;
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        w[j] = (uint32_t)cvt_out<BF16>(o[2 * j] * inv) | ((uint32_t)cvt_out<BF16>(o[2 * j + 1] * inv) << 16);
    *(u32x4 *)orow = w;
}

}  // namespace mi_sgl
