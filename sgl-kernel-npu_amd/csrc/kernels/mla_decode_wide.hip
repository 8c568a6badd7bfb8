// Paged MLA decode, wide variant (see mla_decode.hip for the shared scheme and the reference it replaces:
// python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:5-230).
// Register plan: the compiler's MFMA selection writes every MFMA result to AGPRs once a kernel needs them, and the 256
// fp32 output accumulators already fill that half of the register file -- so the S^T = K.Q^T MFMAs are emitted through
// inline asm with a VGPR destination (16 registers), next to the 144 VGPRs of resident Q^T.  The hazards the compiler
// cannot see for them are covered by hand: dependent MFMAs on one accumulator issue back to back (hardware interlock),
// and 18 wait states separate the last one from the first VALU read of S^T.
#include "device_once.h"
#include "mi_sgl_kernels.h"
#include "mla_common.h"


#ifndef MLAW_CP
#define MLAW_CP ""          // cache-policy suffix of the KV LDS-DMA; " nt" / " sc1" / " sc0 sc1" measured: no effect at C4
#endif

namespace mi_sgl {

// ---------------------------------------------------------------------------------------------------------------------
// Wide variant: kv groups of more than 64 heads (DeepSeek DP-attention decode: 128 heads on one latent head, BASELINE C4).
// One workgroup = ALL 128 heads of a (sequence, KV split): 4 waves on v_mfma_f32_32x32x16 -- twice the FLOPs per
// LDS operand byte and per issue slot of the 16x16x32 form, and each KV tile enters a CU's LDS once for 128 heads instead
// of once per 64 (the 64-head kernel above relies on L2 for its second reader; here L2->LDS traffic = HBM traffic).
// The two contractions split the work differently: QK^T and the softmax by HEAD (wave w: heads 32 w ..+32,
// all 576 dims, Q^T in registers), P.V by OUTPUT DIMENSION (wave w: dims 128 w ..+128 of all 128 heads).  P^T (bf16, 8 KB
// per tile) crosses between the two through an LDS exchange buffer; in return a V tile is read from LDS once per workgroup
// instead of once per wave (24 LDS reads per wave and tile instead of 64 -- with one wave per SIMD a 64-bit LDS read costs
// ~25 issue cycles, and the head-split P.V ran at 52 cycles per MFMA because of it).  Tile pipeline per wave:
//   barrier A | publish P^T(t) | QK^T(t+1) (+ DMA of tile t+3) | barrier B | P(t).V(t) with the softmax of tile t+1 in the
//   shadow of its 32 independent MFMAs                                                    (2 tiles of DMA in flight).
// Measured per tile and wave (shader clocks, C4): barrier A + wait 455, QK^T 2250, barrier B + P.V 1390 -- 4.1k against 4.3k
// for the earlier head-split P.V (3 tiles in flight, no exchange, softmax exposed between QK^T and P.V).
//  * registers: 16 accumulator blocks x 16 = 256 AGPRs, Q^T resident in 144 VGPRs; nothing is staged through registers:
//    tiles of 32 keys arrive by LDS-DMA, one 1-KiB piece per 4 QK k-steps, into a ring of 4 LDS slots (3 tiles = 111 KB in
//    flight per CU).  Block-table entries travel the same way (4-byte LDS-DMA into a small per-wave ring), so no vector
//    load result is ever waited on: the only vmcnt wait is the explicit one at the top of a tile, which leaves the two
//    youngest tiles in flight;
//  * layouts (lane = (c32 = lane & 31, kg = lane >> 5)): S^T[key, head] = K.Q^T with A = K rows (lane: key c32, dims
//    16 ks + 8 kg ..+8), B = Q^T (lane: head c32, same dims), C regs r -> key 8 (r>>2) + 4 kg + (r&3); O^T[d, head] += V^T.P^T
//    with B = P^T = the packed C registers (through the exchange buffer, same lane index) and A = V^T via ds_read_b64_tr_b16.  MFMA row m of output block db
//    is d = 128 (db>>2) + 64 (m>>4) + 16 (db&3) + (m&15), which makes the 32-lane tr-read footprint (4 keys x 2 x 32 B)
//    tile the 64 banks under the 1056-B row stride; 16-B chunks of rows 8..15 and 24..31 are swapped pairwise (applied on
//    the DMA source side) so the 16-lane ds_read_b128 groups of the QK operand are conflict-free too.
// ---------------------------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kT2 = kWideTile, kSlots = 4;
constexpr int kSlotBytes = kT2 * kNopeStride + kT2 * kRopeStride;      // 37888
#ifndef MLAW_QK_AHEAD
#define MLAW_QK_AHEAD 6      // K operand fragments in flight (more does not pay: the chain is not waiting on them)
#endif
constexpr int kRingEntries = 32;                         // block ids per (wave, slot): lanes 32..63 mirror 0..31
constexpr int kRingBytes = 4 * kSlots * kRingEntries * 4;              // per-wave block-table rings
constexpr int kPOff = kSlots * kSlotBytes + kRingBytes;                // P^T exchange buffer [head block 4][k-step 2][lane 64] x 16 B
constexpr int kPBytes = 8192;
constexpr int kFlagOff = kPOff + kPBytes;                              // restart flag (4 B)
constexpr int kWideLds = kFlagOff + 16;                                // 161808
static_assert(kWideLds <= 160 * 1024, "LDS budget");

// S^T chain in VGPRs (see the register plan above)
template <bool BF16>
__device__ __forceinline__ void mfma32_first(f32x16 &d, s16x8 a, s16x8 b)
{
    if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <bool BF16>
__device__ __forceinline__ void mfma32_acc(f32x16 &d, s16x8 a, s16x8 b)
{
    if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_settle(f32x16 &d)      // XDL write -> VALU read of the 16-pass result
{
    asm volatile("s_nop 15\n\ts_nop 2" : "+v"(d));
}

template <bool BF16>
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// value of the lower half-wave's lane (l & 31) and of the upper half-wave's, in every lane: one v_permlane32_swap
struct HalfPair {
    float lo, hi;
};
__device__ __forceinline__ HalfPair halves(float x)
{
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    const unsigned u = __float_as_uint(x);
    // vdst' = {lower lanes of vdst | lower lanes of vsrc}, vsrc' = {upper lanes of vdst | upper lanes of vsrc}
    const u32x2v r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return HalfPair{__uint_as_float(r[0]), __uint_as_float(r[1])};
}

struct WideCtx {
    const MlaParams *p;
    int b, kvh, seq_len, wave, lane;
    uint8_t *lds;
    uint32_t *ring;        // this wave's [kSlots][64] block ids
    uint32_t lds_base, ring_addr;      // LDS byte addresses of lds / ring (M0 values for the DMA)
    int page_shift;                    // log2(page_size) when it is a power of two, else -1 (integer division)
    // cache addressing in 32-bit pieces (the launcher sends caches whose strides do not fit to the 64-head kernel): the general
    // int64 form cost 14 quarter-rate multiplications per tile, in front of the first MFMA of the tile
    const uint16_t *kn_base, *kr_base; // k_nope / k_rope + kv head offset
    uint32_t kn_sblk, kn_srow, kr_sblk, kr_srow;
};

// key of `tile` owned by this lane (lanes 32..63 mirror 0..31), clamped into the sequence
__device__ __forceinline__ int wide_page(const WideCtx &c, int n)
{
    return c.page_shift >= 0 ? n >> c.page_shift : n / c.p->page_size;
}

__device__ __forceinline__ int wide_key(const WideCtx &c, int tile)
{
    int n = tile * kT2 + (c.lane & 31);
    n = n < c.seq_len ? n : c.seq_len - 1;
    return n < 0 ? 0 : n;
}

// Device-scope (sc1) accesses for the pair hand-off of the in-kernel merge (round 4): a write-through store is visible to a reader on any
// XCD once the writer's vmcnt has drained -- no L2 write-back, no L2 invalidate.  Round 2's version published the partial with an
// agent-scope release and read it behind an acquire fence and one load at a time: 202 us against 190-195 us for the separate merge launch.
__device__ __forceinline__ void w4_st_sc1_x4(void *ptr, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void w4_ld_sc1_x4_batch8(f32x4 (&q)[8], const float *const (&a)[8])      // eight independent loads, then ONE wait
{
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\t"
        "global_load_dwordx4 %3, %11, off sc1\n\tglobal_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
        "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
        : "memory");
}
__device__ __forceinline__ float w4_ld_sc1_f32(const float *ptr)
{
    return __uint_as_float(__hip_atomic_load((const uint32_t *)ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void w4_st_sc1_f32(float *ptr, float v)
{
    __hip_atomic_store((uint32_t *)ptr, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS-DMA issued through inline asm: the compiler then sees no vector-memory operation in the tile loop and places no
// vmcnt wait of its own (with the builtin it put an `s_waitcnt vmcnt(0)` in front of the first transpose read of every
// tile, i.e. the whole fill latency sat between QK and PV).  Ordering is explicit instead: one `s_waitcnt vmcnt(10)` per
// tile.  M0 carries the (wave-uniform) LDS destination; one wait state separates its write from the load.
__device__ __forceinline__ uint32_t lds_addr(const void *generic)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)generic;
}
__device__ __forceinline__ void dma16_sbase(uint32_t dst, const void *sbase, uint32_t voff)      // lane l: 16 B from sbase + voff -> dst + 16 l
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MLAW_CP ::"s"(dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ void dma16_vaddr(uint32_t dst, const void *vaddr)                      // lane l: 16 B from its own address
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" MLAW_CP ::"s"(dst), "v"(vaddr) : "memory", "m0");
}
__device__ __forceinline__ void dma4_vaddr(uint32_t dst, const void *vaddr)                       // lane l: 4 B -> dst + 4 l
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(vaddr) : "memory", "m0");
}

// block-table entry of this lane's key of `tile` -> ring
__device__ __forceinline__ void wide_issue_rows(const WideCtx &c, int tile)
{
    const int page = wide_page(c, wide_key(c, tile));
    const int32_t *src = c.p->block_table + (int64_t)c.b * c.p->bt_stride + page;
    if (c.lane < kRingEntries) dma4_vaddr(c.ring_addr + (uint32_t)((tile & (kSlots - 1)) * kRingEntries * 4), src);
}

__device__ __forceinline__ TileRows wide_rows(const WideCtx &c, int tile)
{
    const int n = wide_key(c, tile);
    const int page = wide_page(c, n);
    const uint32_t row = c.page_shift >= 0 ? (uint32_t)n & (uint32_t)(c.p->page_size - 1) : (uint32_t)(n - page * c.p->page_size);
    const uint32_t blk = c.ring[(tile & (kSlots - 1)) * kRingEntries + (c.lane & (kRingEntries - 1))];
    TileRows r;                        // one v_mad_u64_u32 + one 24-bit multiplication each (row < page_size, strides < 2^24: launcher)
    r.nope = (int64_t)((uint64_t)blk * c.kn_sblk + __umul24(row, c.kn_srow));
    r.rope = (int64_t)((uint64_t)blk * c.kr_sblk + __umul24(row, c.kr_srow));
    return r;
}

// Source addresses of this wave's nine DMA pieces of a tile, formed once at the top of the tile (MI_MLAW_PIECES: the eight K-row
// addresses are wave-uniform SGPR pairs, the rope piece keeps a per-lane pointer) instead of two v_readlane + a 64-bit add per piece --
// and, for the rope piece, two ds_bpermute round trips -- in the middle of the QK^T MFMA chain.
struct WidePieces {
    const uint16_t *nope[8];
    const uint16_t *rope;
};
__device__ __forceinline__ WidePieces wide_pieces(const WideCtx &c, const TileRows &rows)
{
    WidePieces q;
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) {
        const int i = c.wave + 4 * idx;
        const int lo = __builtin_amdgcn_readlane((int)(rows.nope & 0xFFFFFFFFll), i);
        const int hi = __builtin_amdgcn_readlane((int)(rows.nope >> 32), i);
        q.nope[idx] = c.kn_base + (((int64_t)hi << 32) | (uint32_t)lo);
    }
    const int key = c.wave * 8 + (c.lane >> 3);
    const int chunk = (c.lane & 7) ^ (key & 7);
    q.rope = c.kr_base + lane_i64(rows.rope, key) + chunk * 8;
    return q;
}
// piece idx 0..7: nope row wave + 4 idx (1 KiB); idx 8: rope rows 8 wave .. +8 (8 x 128 B); `slot` = LDS byte address
__device__ __forceinline__ void wide_issue_piece(const WideCtx &c, const WidePieces &q, uint32_t slot, int idx)
{
    if (idx < 8) {
        const int i = c.wave + 4 * idx;
        const int sw = (i >> 3) & 1;                       // rows 8..15, 24..31: 16-B chunk pairs swapped
        dma16_sbase(slot + (uint32_t)(i * kNopeStride), q.nope[idx], (uint32_t)((c.lane ^ sw) * 16));
    } else {
        dma16_vaddr(slot + (uint32_t)(kT2 * kNopeStride + c.wave * 8 * kRopeStride), q.rope);
    }
}

template <bool BF16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mla_decode_wide_kernel(MlaParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
#ifdef MLAW_TIMING
    const uint64_t t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c32 = lane & 31, kg = lane >> 5;
    const int head_blocks = (p.group + 127) / 128;
    // workgroups b % 8 run on XCD b % 8: the num_splits workgroups of a (sequence, kv head) share an XCD, so the partials they
    // hand each other (in-kernel merge) or to the merge kernel are still in that XCD's L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int u = j / head_blocks, hblk = j % head_blocks;
    const int split = u % p.num_splits;
    const int seq = (u / p.num_splits) * 8 + xcd;              // (b, kvh) pair
    if (seq >= p.batch * p.kv_heads) return;
    const int kvh = seq % p.kv_heads;
    const int b = seq / p.kv_heads;
    const int seq_len = p.seq_lens[b];
    const int ntiles = (seq_len + kT2 - 1) / kT2;
    const int tps = (ntiles + p.num_splits - 1) / p.num_splits;
    const int t_begin = split * tps;
    const int t_end = min(ntiles, t_begin + tps);
    const int hg = hblk * 128 + wave * 32 + c32;
    const bool head_ok = hg < p.group;
    const bool wave_active = hblk * 128 + wave * 32 < p.group;       // wave-uniform; idle waves still feed the DMA
    const int head = kvh * p.group + hg;
    WideCtx cx{&p, b, kvh, seq_len, wave, lane, lds, (uint32_t *)(lds + kSlots * kSlotBytes) + wave * kSlots * kRingEntries, 0, 0,
               (p.page_size & (p.page_size - 1)) == 0 ? __builtin_ctz(p.page_size) : -1,
               p.k_nope + (int64_t)kvh * p.kn_sh, p.k_rope + (int64_t)kvh * p.kr_sh,
               (uint32_t)p.kn_sblk, (uint32_t)p.kn_srow, (uint32_t)p.kr_sblk, (uint32_t)p.kr_srow};
    cx.lds_base = __builtin_amdgcn_readfirstlane(lds_addr(lds));
    cx.ring_addr = cx.lds_base + (uint32_t)(kSlots * kSlotBytes + wave * kSlots * kRingEntries * 4);

    // Block ids of the first kLead tiles are requested BEFORE the 36 Q^T loads: the prologue below then only has to wait for
    // these (older) operations before it can start the KV fill, which runs while Q^T is still in flight.  (All 256 workgroups
    // are in their prologue at once and HBM serves that burst slowly: Q^T, then block ids, then KV one after the other cost
    // 35k cycles per workgroup, 11 % of the kernel at C4.)
    constexpr int kLead = 2;
#pragma unroll
    for (int d = 0; d < kLead; ++d) wide_issue_rows(cx, t_begin + d);
    // Q^T fragments: lane (c32, kg) holds q[head][16 ks + 8 kg .. +8]
    s16x8 qf[36];
    {
        const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)(head_ok ? head : 0) * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
            if (head_ok) qf[ks] = *(const s16x8 *)(qrow + ks * 16 + kg * 8);
            else qf[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    volatile uint32_t *flag = (volatile uint32_t *)(lds + kFlagOff);
    if (threadIdx.x == 0) *(uint32_t *)(lds + kFlagOff) = 0;      // plain LDS store (the volatile one is a flat store + vmcnt(0))
    const float cs = p.sm_scale * 1.4426950408889634f;
    // kLead = how many tiles ahead of the one entering QK^T the fill is issued.  The slot ring has 4 entries and the loop still
    // reads tile t-1 for P.V while tile t is in QK^T, so the fill runs 2 ahead.  Issue order per tile x: R(x + kLead + 2) (block
    // ids), D(x + kLead) (9 pieces) = 10 vector-memory operations; the wait at the top of tile x leaves the youngest 10 in
    // flight, i.e. this wave's pieces of tile x and the block ids of tile x + kLead have landed.  After the barrier tile x
    // is complete in LDS and the slot of tile x-2 is free for tile x+2.
    auto tile_top = [&](int t) -> WidePieces {
#ifndef MLAW_NO_PIECES
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
#endif
        __syncthreads();
        wide_issue_rows(cx, t + kLead + 2);
        return wide_pieces(cx, wide_rows(cx, t + kLead));
    };
    // prologue in steady-state issue order: ... D(t) | R(t + kLead) D(t+1) | ...
    auto prologue = [&]() {
        // the block ids (issued first) have landed once at most the 36 Q^T loads of this wave are outstanding; a wave without
        // heads issued none
        if (__any(head_ok)) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < kLead; ++d) {
            if (d + 2 >= kLead) wide_issue_rows(cx, t_begin + d + 2);      // R(x + kLead + 2) of the virtual iteration x = t_begin + d - kLead
            const WidePieces r = wide_pieces(cx, wide_rows(cx, t_begin + d));
#pragma unroll
            for (int i = 0; i < 9; ++i) wide_issue_piece(cx, r, cx.lds_base + (uint32_t)(((t_begin + d) & (kSlots - 1)) * kSlotBytes), i);
        }
    };
    if (t_begin < t_end) prologue();
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): Q^T is resident (and the first fills have landed): no
                                                               // compiler-visible vector load is pending when the tile loop starts
    // ---- O^T[d, head] += V^T . P^T, split by OUTPUT DIMENSION: wave w owns d in [128 w, 128 w + 128) for all 128 heads of the
    // workgroup (accumulator block dbl*4 + hb = 32 dims x 32 heads), so a V tile is read from LDS once per workgroup instead
    // of once per wave: 16 ds_read_b64_tr_b16 + 8 ds_read_b128 per wave and tile instead of 64 transpose reads.  With one
    // wave per SIMD a 64-bit LDS read costs ~25 issue cycles, which made the head-split P.V LDS-issue-bound (52 cycles per
    // MFMA).  P^T of the other head blocks comes through the exchange buffer: lane l of wave hb stores its two packed
    // B-operand fragments at [hb][kk][l]; after a barrier every wave reads all eight.
    uint8_t *const pbuf = lds + kPOff;
    auto pv_x = [&](int t, auto &&embed) {                     // starts with barrier B
        const uint8_t *buf = lds + (t & (kSlots - 1)) * kSlotBytes;
        const uint8_t *pb = pbuf + lane * 16;
        const int c16 = lane & 15, q16 = (lane >> 4) & 1;
        // rows 8..15 / 24..31 (the `hi` halves) have their 16-B chunk pairs swapped: XOR 16 on the lane's byte offset
        const uint8_t *vlo = buf + (4 * kg + (c16 >> 2)) * kNopeStride + wave * 256 + q16 * 128 + (c16 & 3) * 8;
        const uint8_t *vhi = buf + (4 * kg + (c16 >> 2) + 8) * kNopeStride + wave * 256 + q16 * 128 + (((c16 & 3) * 8) ^ 16);
        auto lda = [&](int step) -> s16x8 {                    // step = kk * 4 + dbl
            const int off = (step >> 2) * 16 * kNopeStride + (step & 3) * 32;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vlo + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vhi + off));
            return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto ldp = [&](int hb, int kk) -> s16x8 { return *(const s16x8 *)(pb + (hb * 2 + kk) * 1024); };
        s16x8 pfr[4], af[3];
        __builtin_amdgcn_sched_barrier(0);
        af[0] = lda(0);                                        // V(t) has been resident since barrier A: its first fragments
        af[1] = lda(1);                                        // are requested BEFORE barrier B and land while the wave waits there
        __builtin_amdgcn_sched_barrier(0);
        // barrier B.  LDS operations of a wave complete in order, so "all but the 4 youngest" covers this wave's P^T stores.
        asm volatile("s_waitcnt lgkmcnt(4)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) pfr[hb] = ldp(hb, 0);
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            __builtin_amdgcn_sched_barrier(0);
            if (step + 2 < 8) af[(step + 2) % 3] = lda(step + 2);
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                const int a = (step & 3) * 4 + hb;
                acc[a] = mfma32<BF16>(af[step % 3], pfr[hb], acc[a]);
                if (step == 3) pfr[hb] = ldp(hb, 1);           // k-step 1 fragments, four MFMAs ahead of their first use
            }
            embed(step);                                       // softmax of the NEXT tile: VALU in the shadow of these independent MFMAs
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto no_embed = [](int) {};
    auto fill_only = [&](int t) {
        const WidePieces rows3 = tile_top(t);
        const uint32_t nslot = cx.lds_base + (uint32_t)(((t + kLead) & (kSlots - 1)) * kSlotBytes);
#pragma unroll
        for (int i = 0; i < 9; ++i) wide_issue_piece(cx, rows3, nslot, i);
    };
    if (!wave_active) {                                        // no heads of its own: P = 0 for its block, DMA share and P.V slice as usual
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) *(u32x4 *)(pbuf + ((wave * 2 + kk) * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t_begin < t_end) {
            fill_only(t_begin);
            for (int t = t_begin; t + 1 < t_end; ++t) {
                fill_only(t + 1);
                pv_x(t, no_embed);
            }
            __syncthreads();
            pv_x(t_end - 1, no_embed);
        }
    }

    // ---- S^T[key, head] = K . Q^T : 36 k-steps of 16 dims, operand ring kAhead deep, one DMA piece per 4 k-steps;
    // returns the tile maximum per head in the scaled log2 domain
    auto qk = [&](int t, const WidePieces &rows3, f32x16 &s) -> float {
        const uint8_t *buf = lds + (t & (kSlots - 1)) * kSlotBytes;
        const uint32_t nslot = cx.lds_base + (uint32_t)(((t + kLead) & (kSlots - 1)) * kSlotBytes);
        // (2 ks + kg) ^ sw == 2 ks + (kg ^ sw): the swizzle folds into the lane base, k-steps are immediate offsets
        const uint8_t *abase = buf + c32 * kNopeStride + ((kg ^ ((c32 >> 3) & 1)) << 4);
        const uint8_t *rbase = buf + kT2 * kNopeStride + c32 * kRopeStride;
        auto lda = [&](int ks) -> s16x8 {
            if (ks < 32) return *(const s16x8 *)(abase + ks * 32);
            return *(const s16x8 *)(rbase + ((((ks - 32) * 2 + kg) ^ (c32 & 7)) << 4));
        };
        constexpr int kAhead = MLAW_QK_AHEAD, kRing = kAhead + 1;
        s16x8 af[kRing];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pre = 0; pre < kAhead; ++pre) af[pre] = lda(pre);
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            if (ks + kAhead < 36) af[(ks + kAhead) % kRing] = lda(ks + kAhead);
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) mfma32_first<BF16>(s, af[0], qf[0]);
            else mfma32_acc<BF16>(s, af[ks % kRing], qf[ks]);
#ifndef MLAW_NO_PIECES
            if ((ks & 3) == 0) wide_issue_piece(cx, rows3, nslot, ks >> 2);
#endif
        }
        mfma32_settle(s);
        __builtin_amdgcn_sched_barrier(0);
        // lane owns head c32 and keys 8 (r>>2) + 4 kg + (r&3); only the tile that crosses seq_len needs the mask
        if ((t + 1) * kT2 > seq_len) {
            asm volatile("" ::: "memory");                     // keep this a real (wave-uniform) branch: if-converted it costs
                                                               // ~40 VALU ops on every tile instead of only the last one
            const int kbase = t * kT2 + 4 * kg;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + 8 * (r >> 2) + (r & 3) >= seq_len) s[r] = -INFINITY;
        }
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        {   // the two half-waves meet through v_permlane32_swap (VALU) instead of ds_bpermute: an LDS round trip at the end of the chain
            const auto sw = halves(tmax);
            tmax = fmaxf(sw.lo, sw.hi);
        }
        return tmax * cs;                                      // sm_scale > 0: max commutes with the scaling
    };

    // One softmax reference per head for the WHOLE key range of this workgroup: the first tile's maximum.  No accumulator
    // is ever rescaled, so the tile loop contains no VALU access to the 256 accumulator registers (with a conditional
    // rescale in it the register allocator moved them to VGPRs and spilled Q^T).  bf16 P spans the fp32 exponent range: a
    // later tile may exceed the reference by 2^64 before anything is at risk; fp16 P must stay below 2^15.  If a tile
    // maximum exceeds the reference by more than kGuard (log2 domain) the sequence is flagged and recomputed by the slow path of
    // the merge kernel that follows this launch (mla_decode.hip: mla_recompute_head; data-dependent and rare).  out = acc / l is
    // invariant to the reference.
    constexpr float kGuard = BF16 ? 64.0f : 11.0f;
#ifdef MLAW_TIMING
    uint64_t tm[4] = {0, 0, 0, 0}, c0, c1;
#define MLAW_TICK(i) c1 = __builtin_amdgcn_s_memtime(); tm[i] += c1 - c0; c0 = c1;
#else
#define MLAW_TICK(i)
#endif
    // software pipeline over tiles.  Interval i (between two tile_top barriers): publish P^T(i) (computed in the previous
    // interval, still in registers) to the exchange buffer, QK^T of tile i+1, barrier B, then this wave's dimension slice of
    // P(i).V(i) with the softmax of tile i+1 in the shadow of its 32 independent MFMAs.  (Under the QK^T chain the same VALU
    // work is NOT free: every instruction between two MFMAs on one accumulator breaks their back-to-back issue.)
    // Barrier A (tile_top): tile i+1 landed, everybody is done with V(i-1) and with the exchange buffer; barrier B: P^T(i) complete.
    auto publish = [&](float psum, const uint32_t (&pk)[8]) {
        {
            const auto sw = halves(psum);
            psum = sw.lo + sw.hi;
        }
        l_run += psum;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            *(u32x4 *)(pbuf + ((wave * 2 + kk) * 64 + lane) * 16) = u32x4{pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // stores done here, whatever gets scheduled before barrier B
    };
    if (wave_active && t_begin < t_end) {
        f32x16 s;
        float psum = 0.f, nm = 0.f;
        uint32_t pk[8];
        // the two empty asm statements pin a piece between the MFMA groups it is written next to (the compiler otherwise
        // gathers all sixteen exponentials after the last MFMA)
        auto piece = [&](int i) {
            float a0 = s[2 * i], a1 = s[2 * i + 1];
            asm volatile("" : "+v"(a0), "+v"(a1));
            const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(a0, cs, nm));
            const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(a1, cs, nm));
            psum += e0 + e1;
            pk[i] = pack2<BF16>(e0, e1);
            asm volatile("" : "+v"(pk[i]), "+v"(psum));
        };
        {
            const WidePieces rows = tile_top(t_begin);
            const float tmax = qk(t_begin, rows, s);
            m_run = tmax;
            nm = (m_run == -INFINITY) ? 0.f : -m_run;
#pragma unroll
            for (int i = 0; i < 8; ++i) piece(i);
        }
        for (int t = t_begin; t + 1 < t_end; ++t) {
#ifdef MLAW_TIMING
            c0 = __builtin_amdgcn_s_memtime();
#endif
            const WidePieces rows = tile_top(t + 1);
            MLAW_TICK(0)
            publish(psum, pk);
            psum = 0.f;
            const float tmax = qk(t + 1, rows, s);
            if (__any(tmax > m_run + kGuard)) *(uint32_t *)(lds + kFlagOff) = 1;
            MLAW_TICK(1)
            pv_x(t, piece);
            MLAW_TICK(2)
        }
        __syncthreads();
        publish(psum, pk);
        pv_x(t_end - 1, no_embed);
    }
#ifdef MLAW_TIMING
    const uint64_t t_loop_end = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x < 64) {
        float *dbg = (float *)p.fix_flags + 1024 + (blockIdx.x * 4 + wave) * 4;
        for (int i = 0; i < 3; ++i) dbg[i] = (float)tm[i] / (float)(t_end - t_begin);
        float *dbg2 = (float *)p.fix_flags + 1024 + 1024 + (blockIdx.x * 4 + wave) * 4;      // whole-kernel phases
        dbg2[0] = (float)(t_loop_end - t_entry);               // entry -> end of the tile loop (prologue + all tiles)
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // fills issued past the last tile
    __syncthreads();
    if (threadIdx.x == 0 && *flag != 0)      // (device-scope store: the pair's second workgroup may read it during this launch)
        __hip_atomic_store(p.fix_flags + b * p.kv_heads + kvh, p.fix_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- epilogue: wave w holds acc[dbl*4 + hb][4 rg + i] = O^T[d][head 32 hb + c32], d = 128 w + 64 (rg>>1) + 16 dbl + 8 (rg&1) + 4 kg + i;
    // the softmax statistics of a head live in the wave that owns it and reach the others through LDS
    float *lmb = (float *)pbuf;                                // [0..127] l, [128..255] m (all P.V reads are behind the barrier above)
    if (kg == 0) {
        lmb[wave * 32 + c32] = wave_active ? l_run : 0.f;
        lmb[128 + wave * 32 + c32] = wave_active ? m_run : -INFINITY;
    }
    // In-kernel merge (num_splits <= 2): the two workgroups of a (sequence, kv head, head block) meet at one word tagged with
    // this call's epoch.  The first to arrive writes its partial, then marks it complete; the second waits for that mark,
    // adds the partner's partial to its own accumulators in registers and writes the final rows -- no merge launch, and half
    // of the partial traffic.  Role 0 = alone (one split).
    enum { kAlone = 0, kFirst = 1, kSecond = 2 };
    uint32_t *const role_lds = (uint32_t *)(lds + kFlagOff + 4);
    uint32_t *const word = p.arrive + ((size_t)b * p.kv_heads + kvh) * head_blocks + hblk;
    const uint32_t tag = p.fix_epoch << 2;
    if (threadIdx.x == 0) {
        uint32_t role = kAlone;
        if (p.inline_merge && p.num_splits == 2) {
            uint32_t old = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (true) {
                if ((old >> 2) == p.fix_epoch) { role = kSecond; break; }
                if (__hip_atomic_compare_exchange_strong(word, &old, tag | 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT)) { role = kFirst; break; }
            }
        }
        *role_lds = role;
    }
    __syncthreads();
    const uint32_t role = p.inline_merge ? *role_lds : (p.num_splits == 1 ? (uint32_t)kAlone : (uint32_t)kFirst);
    const bool finals = role != kFirst;                       // this workgroup writes output rows
    bool flagged = false;
    if (role == kSecond) {
        // ONE lane polls (relaxed); no fence: the partner's partial was stored write-through (sc1) and is read with sc1 loads.  (With an
        // agent-scope release / acquire pair instead -- an L2 write-back and an L2 invalidate per pair -- this lost to the merge launch.)
        if (threadIdx.x == 0) {
            while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (tag | 2u)) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
    }
    if (p.inline_merge && finals)                              // own tiles (LDS flag) or the partner's (hand-off word)
        flagged = *flag != 0 || (role == kSecond && __hip_atomic_load(p.fix_flags + b * p.kv_heads + kvh, __ATOMIC_RELAXED,
                                                                       __HIP_MEMORY_SCOPE_AGENT) == p.fix_epoch);
    if (flagged) {                                            // outgrown softmax reference: exact slow path, one head per wave at a time
        for (int i = 0; i < 32; ++i) {
            const int hg2 = hblk * 128 + wave * 32 + i;
            if (hg2 < p.group) mla_recompute_head<BF16>(p, b, kvh * p.group + hg2, lane);
        }
        return;
    }
    // Rows leave through LDS: written straight from the accumulator layout a wave-store is 64 pieces of 16 bytes in 64
    // different rows; transposed in a wave-private LDS tile, one head block at a time, every half-wave writes 512 (fp32
    // partial) or 256 (output) contiguous bytes of one head.  (The epilogue costs ~21k cycles per workgroup at C4 either way:
    // 256 workgroups x 256 KB of partials is a write burst the memory system takes at ~12 B/clk per CU.)
    constexpr int kEpiRow = 128 * 4 + 16;                      // one head: this wave's 128 dims in fp32, padded
    uint8_t *const tile = lds + wave * (32 * kEpiRow);          // the KV ring is free now
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {                           // static accumulator indices: keep this loop unrolled
        if (hblk * 128 + hb * 32 >= p.group) continue;
#pragma unroll
        for (int dbl = 0; dbl < 4; ++dbl)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = (rg >> 1) * 64 + dbl * 16 + (rg & 1) * 8 + 4 * kg;
                const f32x16 &a = acc[dbl * 4 + hb];
                *(f32x4 *)(tile + c32 * kEpiRow + d * 4) = f32x4{a[4 * rg + 0], a[4 * rg + 1], a[4 * rg + 2], a[4 * rg + 3]};
            }
        // wave-private tile: LDS operations of one wave complete in order, no barrier
        if (!finals) {
            // partial rows (the C4 case): eight rows read from the tile back to back, then their eight stores.  In the general loop
            // below every row is an LDS round trip followed by its store, one after the other, with the wave-uniform `finals` branch
            // splitting each iteration into blocks: 64 serial round trips per wave were most of the epilogue's 21k cycles.
#pragma unroll
            for (int it0 = 0; it0 < 16; it0 += 8) {
                f32x4 o8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o8[j] = *(const f32x4 *)(tile + ((it0 + j) * 2 + (lane >> 5)) * kEpiRow + (lane & 31) * 16);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int hgx = hblk * 128 + hb * 32 + (it0 + j) * 2 + (lane >> 5);
                    const int64_t idx = ((int64_t)b * p.q_heads + kvh * p.group + min(hgx, p.group - 1)) * p.num_splits + split;
                    if (hgx < p.group) {
                        if (p.inline_merge) w4_st_sc1_x4(p.ws_o + idx * kDN + wave * 128 + (lane & 31) * 4, o8[j]);
                        else *(f32x4 *)(p.ws_o + idx * kDN + wave * 128 + (lane & 31) * 4) = o8[j];
                    }
                }
            }
            if (wave == 0 && lane < 32) {                      // softmax statistics of the block's 32 heads: lane = head
                const int hgx = hblk * 128 + hb * 32 + lane;
                if (hgx < p.group) {
                    const int64_t idx = ((int64_t)b * p.q_heads + kvh * p.group + hgx) * p.num_splits + split;
                    w4_st_sc1_f32(p.ws_ml + idx * 2 + 0, lmb[128 + hb * 32 + lane]);
                    w4_st_sc1_f32(p.ws_ml + idx * 2 + 1, lmb[hb * 32 + lane]);
                }
            }
            continue;
        }
        f32x4 qp[16];                                          // second of a pair: the partner's rows, two batches of eight loads
        if (role == kSecond) {
#pragma unroll
            for (int it0 = 0; it0 < 16; it0 += 8) {
                const float *qa[8];
                f32x4 q8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int hgx = min(hblk * 128 + hb * 32 + (it0 + j) * 2 + (lane >> 5), p.group - 1);
                    const int64_t idp = ((int64_t)b * p.q_heads + kvh * p.group + hgx) * p.num_splits + (1 - split);
                    qa[j] = p.ws_o + idp * kDN + wave * 128 + (lane & 31) * 4;
                }
                w4_ld_sc1_x4_batch8(q8, qa);
#pragma unroll
                for (int j = 0; j < 8; ++j) qp[it0 + j] = q8[j];
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int hl = it * 2 + (lane >> 5), ch = lane & 31;
            const int hgx = hblk * 128 + hb * 32 + hl;
            if (hgx >= p.group) continue;
            const int headx = kvh * p.group + hgx;
            const int64_t idx = ((int64_t)b * p.q_heads + headx) * p.num_splits + split;
            f32x4 o = *(const f32x4 *)(tile + hl * kEpiRow + ch * 16);
            const float l_h = lmb[hb * 32 + hl], m_h = lmb[128 + hb * 32 + hl];
            if (finals) {
                // out = (w_s O_s + w_p O_p) / (w_s l_s + w_p l_p), w = exp2(m - max m) (the merge kernel's arithmetic, mla_decode.hip)
                float l_tot = l_h;
                if (role == kSecond) {
                    const int64_t idp = idx - split + (1 - split);
                    const float m_p = w4_ld_sc1_f32(p.ws_ml + idp * 2 + 0), l_p = w4_ld_sc1_f32(p.ws_ml + idp * 2 + 1);
                    const float mx = fmaxf(m_h, m_p);
                    const float w_s = m_h == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_h - mx);
                    const float w_p = m_p == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m_p - mx);
                    l_tot = w_s * l_h + w_p * l_p;
                    const f32x4 q = qp[it];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = w_s * o[e] + w_p * q[e];
                }
                const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
                const uint32_t w0 = (uint32_t)cvt_out<BF16>(o[0] * inv) | ((uint32_t)cvt_out<BF16>(o[1] * inv) << 16);
                const uint32_t w1 = (uint32_t)cvt_out<BF16>(o[2] * inv) | ((uint32_t)cvt_out<BF16>(o[3] * inv) << 16);
                *(uint2 *)(p.out + (int64_t)b * p.o_sb + (int64_t)headx * p.o_sh + wave * 128 + ch * 4) = uint2{w0, w1};
            } else {
                *(f32x4 *)(p.ws_o + idx * kDN + wave * 128 + ch * 4) = o;
                if (wave == 0 && ch == 0) {
                    p.ws_ml[idx * 2 + 0] = m_h;
                    p.ws_ml[idx * 2 + 1] = l_h;
                }
            }
        }
    }
    if (p.inline_merge && role == kFirst) {                    // partial (and the hand-off word, written above) complete -> mark
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have left the CU
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(word, tag | 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the write-through stores have left
    }
#ifdef MLAW_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && blockIdx.x < 64) {
        float *dbg2 = (float *)p.fix_flags + 1024 + 1024 + (blockIdx.x * 4 + wave) * 4;
        dbg2[1] = (float)(__builtin_amdgcn_s_memtime() - t_loop_end);     // epilogue including the store drain
    }
#endif
}


void launch_mla_wide(const MlaParams &p, int dtype, long long units, hipStream_t st)
{
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute((const void *)mla_decode_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);
    }
    const int head_blocks = (p.group + 127) / 128;
    const long long seqs = units / p.num_splits;               // (sequence, kv head) pairs, 8 per grid row of XCDs
    dim3 grid((unsigned)(((seqs + 7) / 8) * 8 * p.num_splits * head_blocks));
    if (dtype == MI_DTYPE_BF16) mla_decode_wide_kernel<true><<<grid, 256, kWideLds, st>>>(p);
    else mla_decode_wide_kernel<false><<<grid, 256, kWideLds, st>>>(p);
}

}  // namespace mi_sgl
