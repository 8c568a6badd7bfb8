// Paged GQA decode for large kv groups: ALL (up to 128) query heads of a kv head in one workgroup, head dims up to (288, 256) -- the
// reference test's own (batch, 128 q heads, 1 kv head, 288 / 256) shape (tests/python/sgl_kernel_npu/test_decode_attention.py:239-244),
// which the generic kernel (gqa_decode.hip: four waves, register-staged 64-key tiles, one tile in flight) ran at 0.15 of HBM.
// Reference replaced: python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:233-450 (same math: fp32 scores, softmax in
// fp32, P rounded to the cache dtype before P.V, out = acc / l).
//
// Structure = mla_decode_wide8s.hip with a separate V tile: eight waves (two per SIMD), 32-key tiles filled by LDS-DMA
// (global_load_lds_dwordx4, one cache row per instruction: 36 lanes of a 288-dim K row, 32 of a 256-dim V row) into a ring of four slots of
// K[32][608 B] + V[32][544 B] (rows padded by 32 B: both operand fetch patterns are conflict-free); a tile lies inside ONE page
// (power-of-two pages of >= 32 keys), so its block id is one scalar load and every row address scalar arithmetic; two tiles in flight.
//   QK^T + softmax by HEAD: wave w owns heads 16 w .. +15; S^T[32 keys, 16 heads] = K . Q^T, 9 k-steps x 2 key blocks on
//     v_mfma_f32_16x16x32 (A = K rows, ds_read_b128; B = Q^T, 36 registers).
//   P . V by OUTPUT DIMENSION: wave w owns 32 of the 256 dims for all 128 heads (4 accumulator blocks of 32 dims x 32 heads on
//     v_mfma_f32_32x32x16, 64 registers); P^T crosses from the head owners to the dimension owners through an 8 KB LDS buffer.
// Per tile and CU: 1088 MFMA cycles against ~3700 cycles of fill at the rate the memory system sustains -- HBM-bound by construction.
// Softmax: ONLINE with a lazy reference.  The reference m of a head moves only when a tile's maximum exceeds it by more than 8 (log2
// domain: P <= 256 fits both cache dtypes and fp32 sums with room to spare); the head's owner then publishes alpha = 2^(m_old - m_new)
// for the dimension owners, who rescale their accumulators behind the tile's second barrier -- for random data that happens in the first
// tiles only, and a tile in which no head moved costs one broadcast LDS read.  Exact in exact arithmetic (any reference cancels in acc / l).
// Measured (tools/probes/gqa_wide_time.py, batch 128 x 4096 keys, 128 heads on one kv head, 288 / 256, bf16; generic kernel beside it): V its
// own cache 150 us = 3.8 TB/s (generic 246), ragged 137 (219); V a view of K 122 us (generic 235), ragged 110 (202).  Built to parity and
// dropped: ONE barrier per tile with P(t - 1) . V(t - 1) and K(t) . Q^T + softmax(t) in the same interval (P^T / alpha / moved double-buffered,
// alpha in the K row pads to stay inside 160 KB) -- all tests green, 131 / 155 us: the tile is not barrier-bound.  With its own V cache the
// kernel sits near what the fill delivers (570 MB); the view form (302 MB) is bound by the tile loop itself (~3300 cycles per 32-key tile:
// QK^T ~800, softmax ~700 on the shared VALU, P.V ~600, barriers ~300, four DMA operations ~100-275 each).
#include "device_once.h"
#include "decode_plan.h"
#include "gqa_wide.h"
#include "mi_sgl_kernels.h"

#ifndef GQAW_DMA_EVERY
#define GQAW_DMA_EVERY (-2)    // a fill operation every n-th QK^T MFMA; -2: spread evenly over the tile's MFMAs (MFMAs / operations: 4 for the view
                               // forms, 2 with a V tile of its own); 0: all in front of the first MFMA; -1: all in the softmax phase (probes)
#endif

namespace mi_gqa_wide {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kLead = 2;
constexpr int KS = kDKP * 2 + 32, VS = kDVP * 2 + 32;                      // 608, 544 bytes per LDS row
// T = keys per tile.  32: four slots of K[32][608 B] + V[32][544 B].  64 (V a column prefix of K, pages of >= 64 keys): three slots of
// K[64][608 B] -- the P.V operand is read from the K tile, and a tile of twice the keys halves what a tile costs besides its MFMAs (two
// barriers, the softmax's latency chain, the list / block-id bookkeeping), which is most of the loop at these head dims: 1088 MFMA cycles
// per 32 keys against ~3300 measured.  Two tiles in flight either way (a fill goes to the slot of the tile before the one being multiplied).
template <int T>
struct Geo {
    static constexpr int kKB = T / 16;                                      // key blocks of 16 (QK^T accumulators per wave, P.V k-steps)
    static constexpr int kSlots = T == 64 ? 3 : 4;
    static constexpr int kSlotBytes = T == 64 ? T * KS : T * (KS + VS);     // 38912 | 36864
    static constexpr int kVOff = T * KS;                                    // V tile inside a slot (T = 32)
    static constexpr int kPxOff = kSlots * kSlotBytes;                      // P^T exchange buffer [head block 4][k-step kKB][lane 64] x 16 B
    static constexpr int kPxBytes = 128 * T * 2;
    static constexpr int kAlphaOff = kPxOff + kPxBytes;                     // float alpha[128]: accumulator rescale of a head in the current tile
    static constexpr int kMovedOff = kAlphaOff + 512;                       // uint32 moved[8]: tile + 1 of the last tile in which wave w moved a reference
    static constexpr int kLds = kMovedOff + 32;                             // 156192 | 133664
    static constexpr int kRows = T / 8;                                     // K (and V) rows a wave fetches per tile
    static_assert(kLds <= 160 * 1024, "LDS budget");
    static_assert((kLead + 1) % kSlots == 0 || kSlots == 4, "the fill of tile t + lead goes to the slot of tile t - 1");
};
constexpr int kQS = kDKP / 32;                                             // 9 k-steps
constexpr float kLazy = 8.0f;

template <bool BF16>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c)
{
    if constexpr (BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool BF16>
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c)
{
    if constexpr (BF16) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    if constexpr (BF16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
    else return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2));
}
// maxima / sums over the four 16-lane rows of a wave (lanes of one head): v_permlane16_swap / v_permlane32_swap, no LDS round trip
__device__ __forceinline__ float max_over_rows(float x)
{
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_over_rows(float x)
{
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// LDS-DMA through inline asm (no compiler-tracked vector memory operation in the tile loop; ordering = the explicit s_waitcnt at the top
// of a tile).  M0 = wave-uniform LDS destination; active lane l: 16 B from sbase + voff -> dst + 16 l.
__device__ __forceinline__ void dma16(uint32_t dst, const void *sbase, uint32_t voff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}
// rows handed to the partner workgroup of a two-piece sequence: write-through stores (visible device-wide behind a vmcnt drain, no release
// fence), read back with loads served past the CU's L1 (no acquire); the compiler does not count these loads: ONE wait names every destination
__device__ __forceinline__ void st_sc1_x4(float *ptr, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld_sc1_x4(const float *ptr)
{
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ void ld_sc1_wait(f32x4 (&r)[8])
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])::"memory");
}
__device__ __forceinline__ void st_agent_f32(float *ptr, float v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent_f32(const float *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Ctx {
    const Params *p;
    int b, seq_len, wave, ntiles, page_shift;
    uint32_t lane16;
    const uint16_t *k_base, *v_base;
};
struct TileAt {
    const uint16_t *k, *v;             // first row of the tile in the K / V cache
    int last;                          // index (0..31) of the tile's last valid key (rows behind it re-read that row: finite bytes under P = 0)
};
__device__ __forceinline__ int tile_clamped(const Ctx &c, int tile) { return min(tile, c.ntiles - 1); }
template <int T>
__device__ __forceinline__ const int32_t *block_id_ptr(const Ctx &c, int tile)
{
    return c.p->block_table + ((int64_t)c.b * c.p->bt_stride + ((tile_clamped(c, tile) * T) >> c.page_shift));
}
// the block id of a tile: a scalar load the compiler does not see (see mla_decode_wide8s.hip: a load it tracks would wait vmcnt(0))
template <int T>
__device__ __forceinline__ int block_id_request(const Ctx &c, int tile)
{
    int v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(block_id_ptr<T>(c, tile)) : "memory");
    return v;
}
__device__ __forceinline__ void block_id_wait(int &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)::"memory"); }
template <int T>
__device__ __forceinline__ TileAt tile_of(const Ctx &c, int tile, int blk)
{
    constexpr int kT = T;
    const int t = tile_clamped(c, tile);
    const uint32_t row0 = (uint32_t)(t * kT) & (uint32_t)(c.p->page_size - 1);
    TileAt r;
    r.k = c.k_base + ((int64_t)blk * c.p->k_sblk + (int64_t)((uint64_t)row0 * (uint64_t)c.p->k_srow));
    r.v = c.v_base + ((int64_t)blk * c.p->v_sblk + (int64_t)((uint64_t)row0 * (uint64_t)c.p->v_srow));
    r.last = min(kT - 1, c.seq_len - 1 - t * kT);
    return r;
}
// operation idx 0 .. T / 8 - 1: K row wave + 8 idx; the next T / 8: V rows (none when V is a column prefix of the K rows: the P.V operand is
// then read from the K tile, like the MLA kernels do).  One cache row per instruction; `slot` = LDS byte address
template <int T>
__device__ __forceinline__ void issue_op(const Ctx &c, const TileAt &tl, uint32_t slot, int idx, bool prologue = false)
{
#ifdef GQAW_NO_DMA           // timing probe: the tile loop without its fill (results are garbage)
    if (!prologue) return;
#endif
    constexpr int kRows = Geo<T>::kRows;
    const int r = c.wave + 8 * (idx % kRows), row = __builtin_amdgcn_readfirstlane(min(r, tl.last));      // (keeps the address arithmetic scalar)
    if (idx < kRows) {
        if (c.lane16 < (uint32_t)c.p->lk * 2u) dma16(slot + (uint32_t)(r * KS), tl.k + (int64_t)row * c.p->k_srow, c.lane16);
    } else {
        if (c.lane16 < (uint32_t)c.p->lv * 2u) dma16(slot + (uint32_t)(Geo<T>::kVOff + r * VS), tl.v + (int64_t)row * c.p->v_srow, c.lane16);
    }
}

template <int N> struct SlotTag { static constexpr int value = N; };

#ifdef GQAW_STAMPS           // timing probe: shader clocks per phase, summed per wave in LDS behind the kernel's own area (tools/probes/gqa_wide_phases.py)
__device__ float g_gqaw_phase[256][8][8];
#define GQAW_TICK(i) do { const uint32_t c1_ = (uint32_t)__builtin_amdgcn_s_memtime(); if (lane == 0) __hip_atomic_fetch_add((uint32_t *)(lds + G::kLds) + wave * 8 + (i), c1_ - c0_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); c0_ = c1_; } while (0)
#else
#define GQAW_TICK(i) do { } while (0)
#endif

// VIEW: the V cache is the first lv columns of the K rows (same pointer and strides -- the reference test builds exactly that,
// test_decode_attention.py:74): one fill serves both GEMMs
// PACK (64-key view form with full-width K rows, lk = 288): a tile's K rows are contiguous in the cache (one page) and 608 bytes apart in the
// slot, so the slot is filled as 38 LINEAR 1 KiB pieces -- lane l of piece j fetches the 16 bytes that belong at LDS byte 1024 j + 16 l
// (row = chunk / 38, column = chunk % 38; the two pad columns of a row re-fetch its last chunk) -- five operations per wave and tile
// instead of eight row operations that use 36 of their 64 lanes.  Tiles that end inside the sequence's last keys take the row form.
template <bool BF16, bool VIEW, int T, bool PACK = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gqa_decode_wide_kernel(Params p)
{
    static_assert(T == 32 || (T == 64 && VIEW), "64-key tiles: V is read from the K tile");
    static_assert(!PACK || (T == 64 && VIEW), "linear fill: the 64-key view form");
    using G = Geo<T>;
    constexpr int kT = T, kKB = G::kKB, kSlots = G::kSlots, kSlotBytes = G::kSlotBytes, kVOff = G::kVOff, kPxOff = G::kPxOff;
    constexpr int kAlphaOff = G::kAlphaOff, kMovedOff = G::kMovedOff;
    constexpr int kRowOps = VIEW ? G::kRows : 2 * G::kRows;     // row form: the K rows (+ the V rows) of a wave
    constexpr int kOps = PACK ? 5 : kRowOps;                    // DMA operations per wave and tile (what the vmcnt arithmetic counts)
    constexpr int VSx = VIEW ? KS : VS, kVBase = VIEW ? 0 : kVOff;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h16 = lane & 15, g = lane >> 4;                  // QK^T / softmax role: head h16 of the wave's 16, key group g
    const int c32 = lane & 31, kg = lane >> 5;                 // P.V role: head c32 of a 32-head block, key half kg
    const int head_blocks = (p.group + 127) / 128;
    // sibling head blocks of a unit share an XCD (workgroup i runs on XCD i mod 8): the second reader of a tile hits that XCD's L2
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int unit = (jj / head_blocks) * 8 + xcd, hblk = jj % head_blocks;
    int split, nsplits, seq, t_begin, t_end;
    if (p.plan) {
        const mi_sgl::PlanItem it = mi_sgl::plan_item(p.plan, (long long)p.batch * p.kv_heads, unit);
        if (unit >= __builtin_amdgcn_readfirstlane(p.plan[0]) || it.seq < 0) return;      // behind the list
        seq = it.seq, split = it.k, nsplits = it.n, t_begin = it.t_begin, t_end = it.t_end;
    } else {
        if (unit >= p.batch * p.kv_heads * p.num_splits) return;
        split = unit % p.num_splits, nsplits = p.num_splits, seq = unit / p.num_splits;
        t_begin = t_end = 0;
    }
    const int kvh = seq % p.kv_heads, b = seq / p.kv_heads;
    const int seq_len = __builtin_amdgcn_readfirstlane(p.seq_lens[b]);
    const int ntiles = (seq_len + kT - 1) / kT;
    if (p.plan) {                                               // clamped to the sequence as it is now; the last piece runs to its end
        t_begin = min(t_begin, ntiles);
        t_end = split == nsplits - 1 ? ntiles : min(t_end, ntiles);
    } else {
        const int tps = (ntiles + p.num_splits - 1) / p.num_splits;
        t_begin = split * tps;
        t_end = min(ntiles, t_begin + tps);
    }
    const int hg = hblk * 128 + wave * 16 + h16;
    const bool head_ok = hg < p.group;
    const bool wave_active = hblk * 128 + wave * 16 < p.group;  // wave-uniform; idle waves still feed the DMA and own a P.V slice
    const int head = kvh * p.group + hg;
    if (__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds) != 0u) __builtin_trap();
    const Ctx cx{&p, b, seq_len, wave, ntiles, __builtin_ctz(p.page_size), (uint32_t)lane * 16u, p.k + (int64_t)kvh * p.k_sh, p.v + (int64_t)kvh * p.v_sh};
    // linear pieces of this wave: j = wave + 8 i (38 pieces; waves 6 and 7 repeat their fourth piece as the fifth: every wave issues kOps)
    uint32_t voffp[PACK ? 5 : 1];
    if constexpr (PACK) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int jj = wave + 8 * i < 38 ? wave + 8 * i : wave + 24;
            const uint32_t c = (uint32_t)(64 * jj + lane), row = (c * 1725u) >> 16, col = c - 38u * row;      // c / 38 exactly for c < 2432
            voffp[i] = row * (uint32_t)(p.k_srow * 2) + min(col, 35u) * 16u;
        }
    }
    // fill operation i (0 .. kOps - 1) of the tile `tl` into the slot at LDS byte `slot`
    auto issue = [&](const TileAt &tl, uint32_t slot, int i, bool prologue = false) {
        if constexpr (PACK) {
            if (tl.last == kT - 1) {
#ifdef GQAW_NO_DMA
                if (!prologue) return;
#endif
                const int jj = wave + 8 * i < 38 ? wave + 8 * i : wave + 24;
                dma16(slot + (uint32_t)(jj * 1024), tl.k, voffp[i]);
            } else if (i < 4) {                                 // a tile that ends in the sequence's last keys: the row form (clamped rows), two per call
                issue_op<T>(cx, tl, slot, 2 * i, prologue);
                issue_op<T>(cx, tl, slot, 2 * i + 1, prologue);
            }
        } else {
            issue_op<T>(cx, tl, slot, i, prologue);
        }
    };

    // head dims below the padded (288, 256): the pad columns of the slots are never a DMA target and must read as zero under the zero tail of
    // Q^T (K) -- a stale NaN times zero is NaN -- and as anything finite under the discarded output dims (V)
    if (p.lk < kDKP || p.lv < kDVP) {
        for (int i = threadIdx.x * 16; i < kSlots * kSlotBytes; i += 512 * 16) *(u32x4 *)(lds + i) = u32x4{0, 0, 0, 0};
        __syncthreads();
    }
    int blk_next = 0;                                           // block id of tile t + lead, t = the tile at whose top it is read
    if (t_begin < t_end) {                                      // prologue: tiles t_begin, t_begin + 1 -> slots 0, 1, before the Q^T loads
        int id0 = block_id_request<T>(cx, t_begin), id1 = block_id_request<T>(cx, t_begin + 1);
        blk_next = block_id_request<T>(cx, t_begin + kLead);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(id0), "+s"(id1), "+s"(blk_next)::"memory");
        const int ids[kLead] = {id0, id1};
#pragma unroll
        for (int d = 0; d < kLead; ++d) {
            const TileAt tl = tile_of<T>(cx, t_begin + d, ids[d]);
#pragma unroll
            for (int i = 0; i < kOps; ++i) issue(tl, (uint32_t)(d * kSlotBytes), i, true);
        }
    }
    // Q^T fragments (B operand of 16x16x32): lane (h16, g) holds q[head][32 ks + 8 g .. +8], zero behind lk
    s16x8 qf[kQS];
    {
        const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)(head_ok ? head : 0) * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < kQS; ++ks) {
            const int d = ks * 32 + g * 8;
            qf[ks] = (head_ok && d < p.lk) ? *(const s16x8 *)(qrow + d) : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_ref = -INFINITY, l_run = 0.f;
    if (threadIdx.x < 8) ((uint32_t *)(lds + kMovedOff))[threadIdx.x] = 0;
    const float cs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.sm_scale * 1.4426950408889634f)));
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): Q^T resident (and the prologue fill): nothing the compiler tracks is pending
    __syncthreads();                                            // moved[] cleared before anybody reads it

    const uint32_t lane16 = cx.lane16;
    // ---- O^T[d, head] += V^T . P^T: wave w owns dims 128 (w >> 2) + 16 (w & 3) + {0..15, 64..79}: MFMA row m = dim 128 (w >> 2) + 64 (m >> 4)
    // + 16 (w & 3) + (m & 15) (the 32-lane transposed fetch then tiles the 64 banks under the 544- and the 608-byte row stride); accumulator block hb =
    // those 32 dims x heads 32 hb .. +31.  Starts behind barrier B.
    const int c16 = lane & 15, q16 = (lane >> 4) & 1;
    const uint32_t v_lane = (uint32_t)(kVBase + (4 * kg + (c16 >> 2)) * VSx + (wave >> 2) * 256 + (wave & 3) * 32 + q16 * 128 + (c16 & 3) * 8);
    // V^T fragment kk of the tile in slot SLOT (keys 16 kk + {4 kg + 0..3, 8 + 4 kg + 0..3}); the tile has been complete since barrier A, so
    // the caller requests all of them AHEAD of barrier B (only P^T, alpha and moved[] need that barrier)
    auto v_frag = [&](auto slot_tag, int kk) -> s16x8 {
        constexpr int SLOT = decltype(slot_tag)::value;
        const uint8_t *vlo = lds + SLOT * kSlotBytes + v_lane;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vlo + kk * 16 * VSx));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vlo + kk * 16 * VSx + 8 * VSx));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto pv = [&](int t, const s16x8 (&a)[kKB]) {
        const uint8_t *pb = lds + kPxOff + opaque(lane16);
        auto ldp = [&](int hb, int kk) -> s16x8 { return *(const s16x8 *)(pb + (hb * kKB + kk) * 1024); };
        // the P^T fragments are requested first; the test for moved references (one broadcast read, wave-uniform) runs while they travel
        s16x8 pf[8];                                           // a ring of eight: fragment i + 8 is requested behind the MFMA that read fragment i
#pragma unroll
        for (int i = 0; i < 8; ++i) pf[i] = ldp(i & 3, i >> 2);
        // references that moved in this tile: rescale the accumulators of those heads before the tile's products are added
        {
            const u32x4 m0 = *(const u32x4 *)(lds + kMovedOff), m1 = *(const u32x4 *)(lds + kMovedOff + 16);
            const uint32_t mv[8] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                // (the two waves of a head block move independently: the alpha entries of the one that did not move are stale)
                const bool lo_moved = __builtin_amdgcn_readfirstlane(mv[2 * hb]) == (uint32_t)(t + 1);
                const bool hi_moved = __builtin_amdgcn_readfirstlane(mv[2 * hb + 1]) == (uint32_t)(t + 1);
                if (lo_moved || hi_moved) {
                    const float al = ((c32 < 16) ? lo_moved : hi_moved) ? ((const float *)(lds + kAlphaOff))[hb * 32 + c32] : 1.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[hb][r] *= al;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4 * kKB; ++i) {
            acc[i & 3] = mfma32<BF16>(a[i >> 2], pf[i & 7], acc[i & 3]);
            if (i + 8 < 4 * kKB) pf[i & 7] = ldp((i + 8) & 3, (i + 8) >> 2);
        }
    };
    // top of tile t: this wave's operations of t have landed (the 8 of tile t + 1 may be in flight), barrier A: tile t complete in LDS,
    // everybody done with tile t - 1 and with the exchange buffer
    auto tile_top = [&](int t) -> TileAt {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kOps * (kLead - 1)) : "memory");
        __syncthreads();
        return tile_of<T>(cx, t + kLead, blk_next);
    };
    auto pv_and_next_id = [&](int t, const s16x8 (&a)[kKB]) {
        int id = block_id_request<T>(cx, t + 1 + kLead);
        pv(t, a);
        block_id_wait(id);
        blk_next = id;
    };
    const int hbw = wave >> 1;                                  // exchange-buffer coordinates of this wave's P^T pieces: consumer lane
    const int lc = (g & 1) * 32 + (wave & 1) * 16 + h16;        // (kg = g & 1, c32 = 16 (w & 1) + h16), half g >> 1 of its 16 bytes
    const uint32_t pdst_off = (uint32_t)(kPxOff + (hbw * kKB * 64 + lc) * 16 + (g >> 1) * 8);      // + 1024 per key block
    const uint32_t a_lane = (uint32_t)(h16 * KS + g * 16);

#ifdef GQAW_STAMPS
    uint32_t c0_ = 0;
    if (lane < 8) ((uint32_t *)(lds + G::kLds))[wave * 8 + lane] = 0;
    const uint64_t clk0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    auto body = [&](auto slot_tag, int t) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr uint32_t nslot = (uint32_t)(((SLOT + kLead) % kSlots) * kSlotBytes);
#ifdef GQAW_STAMPS
        c0_ = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
        const TileAt tl = tile_top(t);
        GQAW_TICK(0);
        if (!wave_active) {                                     // no heads of its own: DMA share, P = 0 (written once, below), P.V slice
#pragma unroll
            for (int i = 0; i < kOps; ++i) issue(tl, nslot, i);
            s16x8 av[kKB];
#pragma unroll
            for (int kk = 0; kk < kKB; ++kk) av[kk] = v_frag(slot_tag, kk);
            asm volatile("s_barrier" ::: "memory");              // barrier B
            pv_and_next_id(t, av);
            return;
        }
        // ---- S^T[key, head] = K . Q^T: 9 k-steps x 2 key blocks of 16, operand fragments three MFMAs ahead, a DMA operation every
        // GQAW_DMA_EVERY MFMAs (0: all of them in front of the first MFMA)
        f32x4 sc[kKB];
#pragma unroll
        for (int kb = 0; kb < kKB; ++kb) sc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const uint8_t *abase = lds + SLOT * kSlotBytes + a_lane;
            auto lda = [&](int step) -> s16x8 { return *(const s16x8 *)(abase + (step % kKB) * 16 * KS + (step / kKB) * 64); };
            constexpr int kAhead = 3, kRing = kAhead + 1, kEvery = GQAW_DMA_EVERY == -2 ? (kKB * kQS) / kOps : GQAW_DMA_EVERY;
            static_assert(kEvery <= 0 || kEvery * kOps <= kKB * kQS, "every operation of a tile is issued inside its QK^T");
            if constexpr (kEvery == 0)
#pragma unroll
                for (int i = 0; i < kOps; ++i) issue(tl, nslot, i);
            s16x8 af[kRing];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pre = 0; pre < kAhead; ++pre) af[pre] = lda(pre);
#pragma unroll
            for (int step = 0; step < kKB * kQS; ++step) {
                __builtin_amdgcn_sched_barrier(0);
                if (step + kAhead < kKB * kQS) af[(step + kAhead) % kRing] = lda(step + kAhead);
                __builtin_amdgcn_sched_barrier(0);
                sc[step % kKB] = mfma16<BF16>(af[step % kRing], qf[step / kKB], sc[step % kKB]);
                if constexpr (kEvery > 0)
                    if (step % kEvery == kEvery - 1 && step / kEvery < kOps) issue(tl, nslot, step / kEvery);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        GQAW_TICK(1);
        // lane (h16, g) holds head h16, keys 16 kb + 4 g + i.  Only the tile that crosses seq_len needs the mask.
        if ((t + 1) * kT > seq_len) {
            const int kbase = t * kT + 4 * g;
#pragma unroll
            for (int kb = 0; kb < kKB; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (kbase + 16 * kb + i >= seq_len) sc[kb][i] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3]));
#pragma unroll
        for (int kb = 1; kb < kKB; ++kb) tmax = fmaxf(tmax, fmaxf(fmaxf(sc[kb][0], sc[kb][1]), fmaxf(sc[kb][2], sc[kb][3])));
        tmax = max_over_rows(tmax) * cs;                        // sm_scale > 0: max commutes with the scaling; a tile below the end holds a key
        const bool move = tmax > m_ref + kLazy || m_ref == -INFINITY;
        if (__any(move)) {
            const float a = move ? (m_ref == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_ref - tmax)) : 1.f;
            if (move) m_ref = tmax;
            l_run *= a;
            if (t != t_begin) {                                 // (first tile: the accumulators are zero, nobody needs alpha)
                if (g == 0) ((float *)(lds + kAlphaOff))[wave * 16 + h16] = a;
                if (lane == 0) ((uint32_t *)(lds + kMovedOff))[wave] = (uint32_t)(t + 1);
            }
        }
        const float nm = -m_ref;
        // scores -> exp2 -> P^T in LDS is the phase's critical path; the running sum follows the LDS writes, the V^T fragments of the P.V
        // phase are requested ahead of barrier B
        uint8_t *pdst = lds + pdst_off;
        float esum[kKB / 2];
#pragma unroll
        for (int kp = 0; kp < kKB; kp += 2) {                   // key blocks in pairs: the sums of a 32-key tile keep their order
            float e[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kp][i], cs, nm));
                e[4 + i] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[kp + 1][i], cs, nm));
            }
            *(uint2 *)(pdst + kp * 1024) = uint2{pack2<BF16>(e[0], e[1]), pack2<BF16>(e[2], e[3])};
            *(uint2 *)(pdst + (kp + 1) * 1024) = uint2{pack2<BF16>(e[4], e[5]), pack2<BF16>(e[6], e[7])};
            esum[kp / 2] = ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
        }
        s16x8 av[kKB];
#pragma unroll
        for (int kk = 0; kk < kKB; ++kk) av[kk] = v_frag(slot_tag, kk);
        if constexpr (GQAW_DMA_EVERY == -1)
#pragma unroll
            for (int i = 0; i < kOps; ++i) issue(tl, nslot, i);
#pragma unroll
        for (int kp = 0; kp < kKB / 2; ++kp) l_run += esum[kp];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GQAW_TICK(2);
        asm volatile("s_barrier" ::: "memory");                   // barrier B: P^T(t), alpha, moved complete
        GQAW_TICK(3);
        pv_and_next_id(t, av);
        GQAW_TICK(4);
    };
    if (!wave_active) {
#pragma unroll
        for (int kb = 0; kb < kKB; ++kb) *(uint2 *)(lds + pdst_off + kb * 1024) = uint2{0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int t = t_begin; t < t_end;) {
        body(SlotTag<0>{}, t);
        if (++t >= t_end) break;
        body(SlotTag<1>{}, t);
        if (++t >= t_end) break;
        body(SlotTag<2>{}, t);
        if constexpr (kSlots == 4) {
            if (++t >= t_end) break;
            body(SlotTag<3>{}, t);
        }
        ++t;
    }
#ifdef GQAW_STAMPS
    if (lane == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 5; ++i) g_gqaw_phase[blockIdx.x][wave][i] = (float)((volatile uint32_t *)(lds + G::kLds))[wave * 8 + i] / (float)max(1, t_end - t_begin);
        g_gqaw_phase[blockIdx.x][wave][5] = (float)(__builtin_amdgcn_s_memtime() - clk0_);
        g_gqaw_phase[blockIdx.x][wave][6] = (float)(__builtin_amdgcn_s_memrealtime() - rt0_);
        g_gqaw_phase[blockIdx.x][wave][7] = (float)(t_end - t_begin);
    }
#endif
    l_run = sum_over_rows(l_run);                               // the four key groups of a head
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // fills issued past the last tile
    __syncthreads();

    // ---- epilogue: acc[hb][4 rq + i] = O^T[d][head 32 hb + c32], d = 128 (w >> 2) + 64 (rq >> 1) + 16 (w & 3) + 8 (rq & 1) + 4 kg + i; the
    // softmax statistics of a head live in the wave that owns it and reach the others through LDS
    float *lmb = (float *)(lds + kPxOff);                      // [0..127] l, [128..255] m
    if (g == 0) {
        lmb[wave * 16 + h16] = wave_active ? l_run : 0.f;
        lmb[128 + wave * 16 + h16] = wave_active ? m_ref : -INFINITY;
    }
    __syncthreads();
    // the lane's roles are derived AGAIN from a thread id the optimiser cannot match with the prologue's: otherwise 4 kg and friends are kept in
    // vector registers across the tile loop, which runs at the register limit (the pair finish below cost two spills inside the loop)
    uint32_t tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, c32_e = tid_e & 31, kg_e = (tid_e >> 5) & 1;
    const int dbase = (wave >> 2) * 128 + (wave & 3) * 16 + 4 * kg_e;
    // the pair words are read from the kernel argument segment HERE (an address the optimiser cannot see through): fetched with the other
    // arguments in the prologue they stay in scalar registers for the whole tile loop, which has none to spare (spills into the loop)
    const __attribute__((address_space(4))) Params *kp = (const __attribute__((address_space(4))) Params *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    uint64_t *const pair_flags = kp->pair_flags, *const need_merge = kp->need_merge;
    const uint64_t pair_tag = kp->pair_tag;
    const bool pair = pair_flags != nullptr && nsplits == 2 && head_blocks == 1;
    // partial row of (head of the group, piece): the work list's items of a sequence are consecutive, piece k at first + k
    auto prow = [&](int hgx, int s) -> int64_t {
        return p.plan ? (int64_t)(unit - split + s) * p.group + hgx : ((int64_t)b * p.q_heads + kvh * p.group + hgx) * nsplits + s;
    };
    auto leave_to_merge_kernel = [&]() {
        if (tid_e == 0 && need_merge) __hip_atomic_store(need_merge, pair_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (!pair) {
        if (nsplits > 1) leave_to_merge_kernel();
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
            const int hgx = hblk * 128 + hb * 32 + c32_e;
            if (hgx >= p.group) continue;
            const int headx = kvh * p.group + hgx;
            if (nsplits == 1) {
                const float l_h = lmb[hb * 32 + c32_e];
                const float inv = l_h > 0.f ? 1.f / l_h : 0.f;
                uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)headx * p.o_sh;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = dbase + (rq >> 1) * 64 + (rq & 1) * 8;
                    if (d >= p.lv) continue;                   // lv % 8 == 0: the 4 dims are in or out together
                    *(uint2 *)(orow + d) = uint2{pack2<BF16>(acc[hb][4 * rq + 0] * inv, acc[hb][4 * rq + 1] * inv),
                                                pack2<BF16>(acc[hb][4 * rq + 2] * inv, acc[hb][4 * rq + 3] * inv)};
                }
            } else {
                const int64_t idx = p.plan ? (int64_t)unit * p.group + hgx : ((int64_t)b * p.q_heads + headx) * p.num_splits + split;
                float *po = p.ws_o + idx * kDVP;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d = dbase + (rq >> 1) * 64 + (rq & 1) * 8;
                    *(f32x4 *)(po + d) = f32x4{acc[hb][4 * rq + 0], acc[hb][4 * rq + 1], acc[hb][4 * rq + 2], acc[hb][4 * rq + 3]};
                }
                if (wave == 0 && kg_e == 0) {
                    p.ws_ml[idx * 2 + 0] = lmb[128 + hb * 32 + c32_e];
                    p.ws_ml[idx * 2 + 1] = lmb[hb * 32 + c32_e];
                }
            }
        }
        return;
    }

    // ---- the pair: a sequence in exactly two pieces (BASELINE-shaped batches: 128 sequences on 256 CUs).  Piece s finishes heads 64 s ..
    // 64 s + 63 of the group.  Export (all eight waves: every wave holds 32 dims of every head): the partner's heads, written through, and the
    // statistics of all heads.  "My rows are visible" -> the partner's word (bounded wait) -> own heads = w0 a0 + w1 a1 in piece order, exactly
    // gqa_merge_kernel's sums, from the own accumulators and the partner's rows.  A partner that does not show up in time (not resident: more
    // items than the chip runs at once) costs nothing but the wait: this workgroup then writes the rows of its own heads as well and leaves
    // them unmarked -- both partials of those heads are in the workspace, and the merge kernel, which skips only marked heads, does the work.
    const int piece = split;
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int hgx = r * 64 + lane_e;
            if (hgx < p.group) {
                st_agent_f32(p.ws_ml + prow(hgx, piece) * 2 + 0, lmb[128 + hgx]);
                st_agent_f32(p.ws_ml + prow(hgx, piece) * 2 + 1, lmb[hgx]);
            }
        }
    }
    auto rows_out = [&](int first_hb, bool through) {           // head blocks first_hb, first_hb + 1 (wave-uniform) of this piece's partial
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
            if ((hb >> 1) != (first_hb >> 1)) continue;
            const int hgx = hb * 32 + c32_e;
            if (hgx >= p.group) continue;
            float *po = p.ws_o + prow(hgx, piece) * kDVP;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int d = dbase + (rq >> 1) * 64 + (rq & 1) * 8;
                const f32x4 v = f32x4{acc[hb][4 * rq + 0], acc[hb][4 * rq + 1], acc[hb][4 * rq + 2], acc[hb][4 * rq + 3]};
                if (through) st_sc1_x4(po + d, v);
                else *(f32x4 *)(po + d) = v;
            }
        }
    };
    rows_out(2 * (piece ^ 1), true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave: its write-through stores have left
    __syncthreads();
    uint32_t *const ok_word = (uint32_t *)(lmb + 256);
    const int64_t fbase = ((int64_t)b * p.kv_heads + kvh) * 2;
    if (tid_e == 0) {
        if (!(kp->pair_withhold && piece == 1))
            __hip_atomic_store(pair_flags + fbase + piece, pair_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t ok = 1;
        while (__hip_atomic_load(pair_flags + fbase + (piece ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != pair_tag) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000ull) {      // 200 us at 100 MHz
                ok = 0;
                break;
            }
        }
        *ok_word = ok;
    }
    __syncthreads();
    if (*(volatile uint32_t *)ok_word == 0) {
        leave_to_merge_kernel();
        rows_out(2 * piece, false);
        return;
    }
    // "this piece finishes its heads": the sign of their sums (nobody reads those words before the merge kernel)
    if (wave == 0) {
        const int hgx = piece * 64 + lane_e;
        if (hgx < p.group) st_agent_f32(p.ws_ml + prow(hgx, piece) * 2 + 1, -lmb[hgx]);
    }
    // the partner's statistics and rows of this lane_e's two heads (head block 2 piece + j, head c32_e), everything in flight before the first use
    float m_par[2], l_par[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int hgx = min(piece * 64 + j * 32 + c32_e, p.group - 1);
        m_par[j] = ld_agent_f32(p.ws_ml + prow(hgx, piece ^ 1) * 2 + 0);
        l_par[j] = ld_agent_f32(p.ws_ml + prow(hgx, piece ^ 1) * 2 + 1);
    }
    f32x4 pr[8];                                                // [j][rq]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int hgx = min(piece * 64 + j * 32 + c32_e, p.group - 1);
        const float *src = p.ws_o + prow(hgx, piece ^ 1) * kDVP;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) pr[j * 4 + rq] = ld_sc1_x4(src + dbase + (rq >> 1) * 64 + (rq & 1) * 8);
    }
    ld_sc1_wait(pr);
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {
        if ((hb >> 1) != piece) continue;
        const int j = hb & 1;
        const int hgx = hb * 32 + c32_e;
        if (hgx >= p.group) continue;
        const float m_own = lmb[128 + hgx], l_own = lmb[hgx];
        const float m0 = piece == 0 ? m_own : m_par[j], m1 = piece == 0 ? m_par[j] : m_own;
        const float l0 = piece == 0 ? l_own : l_par[j], l1 = piece == 0 ? l_par[j] : l_own;
        // gqa_merge_kernel's arithmetic for two pieces, operation by operation
        float M = -INFINITY;
        M = fmaxf(M, m0);
        M = fmaxf(M, m1);
        const float w0 = __builtin_amdgcn_exp2f(m0 - M), w1 = __builtin_amdgcn_exp2f(m1 - M);
        float L = 0.f;
        if (m0 != -INFINITY) L = __builtin_fmaf(w0, l0, L);
        if (m1 != -INFINITY) L = __builtin_fmaf(w1, l1, L);
        const float inv = L > 0.f ? 1.f / L : 0.f;
        uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)(kvh * p.group + hgx) * p.o_sh;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int d = dbase + (rq >> 1) * 64 + (rq & 1) * 8;
            if (d >= p.lv) continue;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float own = acc[hb][4 * rq + i], par = pr[j * 4 + rq][i];
                const float a0 = piece == 0 ? own : par, a1 = piece == 0 ? par : own;
                o[i] = 0.f;
                if (m0 != -INFINITY) o[i] = __builtin_fmaf(w0, a0, o[i]);
                if (m1 != -INFINITY) o[i] = __builtin_fmaf(w1, a1, o[i]);
            }
            *(uint2 *)(orow + d) = uint2{pack2<BF16>(o[0] * inv, o[1] * inv), pack2<BF16>(o[2] * inv, o[3] * inv)};
        }
    }
}

}  // namespace

bool applies(int group, int lk, int lv, int page_size, int64_t k_sblk, int64_t k_srow, int64_t v_sblk, int64_t v_srow)
{
    static const bool allow = !(getenv("MI_GQA_WIDE") && atoi(getenv("MI_GQA_WIDE")) == 0);
    auto fits = [](int64_t v, int bits) { return v >= 0 && v < (1ll << bits); };
    return allow && group > 64 && lk > 192 && lk <= kDKP && lv <= kDVP && (lk % 8) == 0 && (lv % 8) == 0 && (page_size & (page_size - 1)) == 0 &&
           page_size >= kTile && fits(k_sblk, 40) && fits(v_sblk, 40) && fits(k_srow, 31) && fits(v_srow, 31);
}

#ifdef GQAW_STAMPS
extern "C" int mi_gqaw_phases(void *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gqaw_phase), sizeof(float) * 256 * 8 * 8); }
#define GQAW_LDS_EXTRA 256
#else
#define GQAW_LDS_EXTRA 0
#endif
static bool is_view(const Params &p) { return p.v == p.k && p.v_sblk == p.k_sblk && p.v_srow == p.k_srow && p.v_sh == p.k_sh && p.lv <= p.lk; }
// keys per tile of the instance that serves `p` (the unit of its work list): 64 when V is a column prefix of K and a page holds a tile
int tile_keys(const Params &p)
{
    static const bool allow64 = !(getenv("MI_GQA_WIDE_T64") && atoi(getenv("MI_GQA_WIDE_T64")) == 0);
    return allow64 && is_view(p) && p.page_size >= 64 ? 64 : 32;
}

void launch(const Params &p, int dtype, long long units, hipStream_t st)
{
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
#define MI_GQAW_ATTR(B, V, T) (void)hipFuncSetAttribute((const void *)gqa_decode_wide_kernel<B, V, T>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<T>::kLds + GQAW_LDS_EXTRA)
        MI_GQAW_ATTR(true, false, 32); MI_GQAW_ATTR(false, false, 32); MI_GQAW_ATTR(true, true, 32); MI_GQAW_ATTR(false, true, 32);
        MI_GQAW_ATTR(true, true, 64); MI_GQAW_ATTR(false, true, 64);
        (void)hipFuncSetAttribute((const void *)gqa_decode_wide_kernel<true, true, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<64>::kLds + GQAW_LDS_EXTRA);
        (void)hipFuncSetAttribute((const void *)gqa_decode_wide_kernel<false, true, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Geo<64>::kLds + GQAW_LDS_EXTRA);
#undef MI_GQAW_ATTR
    }
    const int head_blocks = (p.group + 127) / 128;
    dim3 grid((unsigned)(((units + 7) / 8) * 8 * head_blocks));
    const bool view = is_view(p);
    const int tile = tile_keys(p);
    // linear fill of the 64-key view form: full-width K rows whose in-tile byte offsets fit the 32-bit lane offset (MI_GQA_WIDE_PACK=0: row form)
    static const bool allow_pack = !(getenv("MI_GQA_WIDE_PACK") && atoi(getenv("MI_GQA_WIDE_PACK")) == 0);
    const bool pack = allow_pack && tile == 64 && p.lk == kDKP && p.k_srow * 2 * 64 < (1ll << 31);
#define MI_GQAW_LAUNCH(B, V, T) gqa_decode_wide_kernel<B, V, T><<<grid, 512, Geo<T>::kLds + GQAW_LDS_EXTRA, st>>>(p)
    if (dtype == MI_DTYPE_BF16) {
        if (tile == 64 && pack) gqa_decode_wide_kernel<true, true, 64, true><<<grid, 512, Geo<64>::kLds + GQAW_LDS_EXTRA, st>>>(p);
        else if (tile == 64) MI_GQAW_LAUNCH(true, true, 64);
        else if (view) MI_GQAW_LAUNCH(true, true, 32);
        else MI_GQAW_LAUNCH(true, false, 32);
    } else {
        if (tile == 64 && pack) gqa_decode_wide_kernel<false, true, 64, true><<<grid, 512, Geo<64>::kLds + GQAW_LDS_EXTRA, st>>>(p);
        else if (tile == 64) MI_GQAW_LAUNCH(false, true, 64);
        else if (view) MI_GQAW_LAUNCH(false, true, 32);
        else MI_GQAW_LAUNCH(false, false, 32);
    }
#undef MI_GQAW_LAUNCH
}

}  // namespace mi_gqa_wide
