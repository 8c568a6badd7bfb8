// Paged MLA decode, eight waves, SCALAR block ids -- kv groups of 65..128 heads on power-of-two pages of >= 32 keys, BASELINE C4.
// Reference replaced: python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:5-230 (numerics contract of mla_decode_wide.hip).
//
// Same work split as mla_decode_wide8.hip (QK^T + softmax by 16-head slices on v_mfma_f32_16x16x32, P.V by 64-dim slices on 32x32x16, P^T
// through an 8 KB exchange buffer, two barriers per 32-key tile) around a different FILL path.  What round 5 measured on that kernel by
// compiling parts out (tools/probes/mla_q_ab.py, C4, one process; us incl. the 16 us merge launch): everything 188; without the KV fill in
// the tile loop 142; fill + softmax + barriers only (no MFMA, no operand read) 149; neither 50.  The two halves add up to 240 and overlap
// to 188: the kernel is bound by its fill and by what the fill costs the compute phases -- not by the QK^T operand latency (a ring 3..7
// deep changed nothing; in isolation the phase runs 1060-1170 cycles at any depth, tools/probes/ubench/lds_mfma_rate.hip).  The fill path
// of the four-slot kernel: per tile and wave one 4-byte LDS-DMA of block ids, a ring read, two 64-bit multiply-adds, eight v_readlane, two
// ds_bpermute and their waits between barrier A and the first MFMA (~400 cycles, serial), and only two tiles in flight although it owns four
// slots.  Here:
//   * a 32-key tile lies inside ONE page (power-of-two pages of >= 32 keys; other page sizes take mla_decode_wide8.hip): its block id is ONE
//     scalar load, requested two tiles before it is used, and every piece address is scalar arithmetic (s_mul / s_add) placed between the
//     MFMAs -- no block-id ring, no readlane, no bpermute, no per-tile vector work at all except the rope piece's clamped row offset;
//   * the lead is a constant (MLA8S_LEAD tiles requested ahead, at most slots - 1).  It does not matter: 1, 2 and 3 tiles ahead run 178 / 176 /
//     177 us, and the fill-only loop 151 / 147 / 148 -- the fill is bound by what the memory system delivers to this access pattern
//     (604 MB in ~117 us of loop = 5.2 TB/s against 6.2 TB/s for the streaming microbenchmark, profiles/r03_cu_fetch_rate.txt), not by latency;
//   * the tile loop is unrolled by the four slots: every LDS address is lane base + compile-time constant, DMA destinations are literals.
// What bounds the loop, round 5 (the stamp probe below, -DMLA8S_STAMPS, and the ablation switches): with the fill nontemporal the fill + softmax
// + barriers alone run 117 us at C4 (6.5 TB/s in the loop), everything but the fill 136 us, both together 173 us -- and the SHADER CLOCK
// under the three is 2.39, 2.07 and 1.84 GHz (s_memtime against the 100 MHz counter over each workgroup's loop): MFMA + LDS operand reads
// + the HBM stream at once put the chip at its power limit, so cycles saved inside the tile come back as a lower clock.  Consistently,
// nothing that only re-times the tile moved the end-to-end time: the fill pieces under QK^T / in the softmax phase / at the head of P.V
// (175-177 us), static priority for the younger wave of each SIMD, a P.V that lags one tile with the softmax of tile u - 1 interleaved
// behind the QK^T MFMAs of tile u (no softmax phase at all: built to full parity -- 338 tests, bit-identical -- 172.3 vs 171.2 us, ragged
// 106.7 vs 104.4 with its extra drain iteration; `git show` of the commit that records this line has the loop).  What would move it is
// energy per tile -- and the LDS operand traffic is not the big part of it: with every second K fragment of QK^T not read at all
// (-DMLA8S_HALF_QK_READS: 288 -> 144 KB of LDS reads per tile and CU, what a k-split QK^T on 32x32x16 would read) the kernel gains 3 %.
// Numerics, softmax reference (first tile's maximum, flagged sequences recomputed by the merge kernel), work list, partial-row layout and
// epilogue: as mla_decode_wide8.hip, same MFMA shapes and summation order (bit-identical results).
#include "device_once.h"
#include "mi_sgl_kernels.h"
#include "mla_common.h"

#ifndef MLA8S_LEAD
#define MLA8S_LEAD 2           // tiles requested ahead of the one being multiplied (<= slots - 1); 1, 2, 3 measured: 178 / 176 / 177 us at C4
#endif
#ifndef MLA8S_DMA_POLICY
#define MLA8S_DMA_POLICY " nt" // cache policy of the KV fill (read once): nontemporal.  One process, alternating, C4: "" 173.9 / 106.2 us (full / ragged),
                               // " nt" 172.3 / 104.8, " sc1" 173.6 / 105.7, " sc0 sc1" 173.8 / 105.7
#endif
#ifndef MLA8S_DMA_PLACE
#define MLA8S_DMA_PLACE 0      // where a tile's five fill pieces are issued: 0 = spread over the QK^T MFMAs, 1 = in the softmax (VALU-only) phase, 2 = at the head of P.V
#endif
#ifndef MLA8S_PRIO
#define MLA8S_PRIO 0           // 1: s_setprio 1 for the younger wave of each SIMD (waves 4..7) for the whole loop
#endif
#ifndef MLA8S_AHEAD
#define MLA8S_AHEAD 2          // K operand fragments in flight in front of the QK^T MFMA that consumes them
#endif

namespace mi_sgl {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kST = kWideTile, kSSlots = 4, kSWaves = 8, kSLead = MLA8S_LEAD;
static_assert(kSLead >= 1 && kSLead <= kSSlots - 1, "a fill goes to the slot of the tile before the one being multiplied");
constexpr int kSSlotBytes = kST * kNopeStride + kST * kRopeStride;         // 37888
constexpr int kPxOff = kSSlots * kSSlotBytes;                              // P^T exchange buffer [head block 4][k-step 2][lane 64] x 16 B
constexpr int kPxBytes = 8192;
constexpr int kSLds = kPxOff + kPxBytes;                                    // 159744
constexpr int kSFlagOff = (kSSlots - 1) * kSSlotBytes + (kST - 1) * kNopeStride + kDN * 2;      // pad of the last K row: never a DMA target
static_assert(kSLds <= 160 * 1024, "LDS budget");

template <bool BF16>
__device__ __forceinline__ void mfma16_first(f32x4 &d, s16x8 a, s16x8 b)
{
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <bool BF16>
__device__ __forceinline__ void mfma16_acc(f32x4 &d, s16x8 a, s16x8 b)
{
    if constexpr (BF16) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_settle(f32x4 &a, f32x4 &b)          // XDL write -> VALU read (8-pass result: 11 wait states)
{
    asm volatile("s_nop 15\n\ts_nop 2" : "+v"(a), "+v"(b));
}
template <bool BF16>
__device__ __forceinline__ f32x16 mfma32(s16x8 a, s16x8 b, f32x16 c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// LDS-DMA through inline asm (the compiler tracks no vector-memory operation in the tile loop and places no vmcnt wait of its own;
// ordering is the explicit s_waitcnt at the top of a tile).  M0 = wave-uniform LDS destination; lane l: 16 B from sbase + voff -> dst + 16 l.
__device__ __forceinline__ void dma16(uint32_t dst, const void *sbase, uint32_t voff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MLA8S_DMA_POLICY ::"s"(dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
// a * s + c with the wave-uniform factor read from its scalar register (the compiler keeps a vector copy of it across the loop otherwise)
__device__ __forceinline__ float fma_s(float a, float s, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(s), "v"(c));
    return r;
}
// 16-byte write-through store (leaves the XCD's L2 at once: another workgroup sees it behind this wave's vmcnt drain, no release fence)
__device__ __forceinline__ void st_sc1_x4(float *ptr, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
// 16-byte loads of rows another workgroup wrote through (sc1: served past this CU's L1): "=v" loads, then ONE wait statement naming
// every destination before the first use (the compiler does not count these loads)
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
__device__ __forceinline__ void st_agent_f32(float *ptr, float v)
{
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t max3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// a per-lane value the optimiser must re-derive where it is used: keeps loop invariants from being hoisted into registers of their own
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// Where the 32 keys of a tile live: ONE page (32 | page size), so one block id and 32 consecutive rows, the last of them clamped to the
// sequence's last key (rows behind it are masked later, but their bytes must be finite: P = 0 times a NaN row is NaN).  All wave-uniform.
struct CtxS {
    const MlaParams *p;
    int b, seq_len, wave, ntiles;
    uint32_t lane16;                   // 16 x lane: the one lane-derived register that lives across the tile loop
    int page_shift;
    const uint16_t *kn_base, *kr_base;
    uint32_t kn_sblk, kn_srow, kr_sblk, kr_srow;
};
struct TileS {
    const uint16_t *kn, *kr;           // first row of the tile in the nope / rope cache (elements)
    int last;                          // index (0..31) of the tile's last valid key
};
__device__ __forceinline__ int tile_clamped(const CtxS &c, int tile) { return min(tile, c.ntiles - 1); }      // fills past the end re-read the last tile
__device__ __forceinline__ const int32_t *block_id_ptr(const CtxS &c, int tile)
{
    return c.p->block_table + ((int64_t)c.b * c.p->bt_stride + ((tile_clamped(c, tile) * kST) >> c.page_shift));
}
// In the tile loop the block id is a scalar load the COMPILER DOES NOT SEE (a load it tracks would be a vector load -- the loop's
// "memory"-clobbering DMA statements rule out s_load for it -- and its wait would be vmcnt(0): every DMA piece in flight).  Requested behind
// barrier B, awaited behind the P.V MFMAs of the same tile (block_id_wait ties the register to the wait, so no use can move in front of
// it).  The compiler's own lgkmcnt waits for LDS reads in between only become more conservative with one more operation outstanding.
__device__ __forceinline__ int block_id_request(const CtxS &c, int tile)
{
    int v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(block_id_ptr(c, tile)) : "memory");
    return v;
}
__device__ __forceinline__ void block_id_wait(int &v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)::"memory"); }
__device__ __forceinline__ TileS tile_of(const CtxS &c, int tile, int blk)
{
    const int t = tile_clamped(c, tile);
    const uint32_t row0 = (uint32_t)(t * kST) & (uint32_t)(c.p->page_size - 1);
    TileS r;
    // 32-bit strides (the launcher sends caches whose strides do not fit to the 64-head kernel); block offsets are 64-bit products
    r.kn = c.kn_base + ((uint64_t)(uint32_t)blk * c.kn_sblk + (uint64_t)(row0 * c.kn_srow));
    r.kr = c.kr_base + ((uint64_t)(uint32_t)blk * c.kr_sblk + (uint64_t)(row0 * c.kr_srow));
    r.last = min(kST - 1, c.seq_len - 1 - t * kST);
    return r;
}
// piece idx 0..3: K row wave + 8 idx (1 KiB); idx 4: rope rows 4 wave .. +3 (4 x 128 B, lanes 0..31; 16-byte chunks swizzled by the row's low
// three bits on the source side).  `slot` = LDS byte address (a literal after inlining)
__device__ __forceinline__ void issue_piece(const CtxS &c, const TileS &tl, uint32_t slot, int idx, bool prologue = false)
{
#ifdef MLA8S_NO_DMA          // timing probe: the tile loop without its KV fill (results are garbage)
    if (!prologue) return;
#endif
    if (idx < 4) {
        const int row = __builtin_amdgcn_readfirstlane(min(c.wave + 8 * idx, tl.last));      // (keeps the address arithmetic scalar)
        dma16(slot + (uint32_t)((c.wave + 8 * idx) * kNopeStride), tl.kn + (uint32_t)row * c.kn_srow, c.lane16);
    } else {
#ifdef MLA8S_NO_ROPE         // timing probe: the fill without its rope pieces (results are garbage)
        if (!prologue) return;
#endif
        const uint32_t lane = opaque(c.lane16) >> 4;
        const uint32_t key = (uint32_t)c.wave * 4u + ((lane >> 3) & 3u);
        const uint32_t chunk = (lane & 7u) ^ (key & 7u);
        const uint32_t voff = (min(key, (uint32_t)tl.last) * c.kr_srow + chunk * 8u) * 2u;
        if (c.lane16 < 32 * 16) dma16(slot + (uint32_t)(kST * kNopeStride + c.wave * 4 * kRopeStride), tl.kr, voff);
    }
}

template <int N> struct SlotTag { static constexpr int value = N; };

#ifdef MLA8S_STAMPS          // timing probe: 100 MHz stamps of the epilogue's steps, one row per workgroup (tools/probes/time_mla_pair.py)
__device__ unsigned long long g_mla8s_stamp[1024][8];
#define MLA8S_STAMP(i) do { if (lane == 0 && blockIdx.x < 1024) g_mla8s_stamp[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ float g_mla8s_phase[256][8][8];     // shader clocks per tile and wave: [own fill wait, barrier A, QK^T, softmax + publish, barrier B, P.V, loop total, tiles]
// (sums kept in LDS behind the exchange buffer: accumulators in registers pushed the kernel into scratch)
#define MLA8S_TICK(i) do { const uint32_t c1_ = (uint32_t)__builtin_amdgcn_s_memtime(); if (lane == 0) __hip_atomic_fetch_add((uint32_t *)(lds + kSLds) + wave * 8 + (i), c1_ - c0_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); c0_ = c1_; } while (0)
#else
#define MLA8S_STAMP(i) do { } while (0)
#define MLA8S_TICK(i) do { } while (0)
#endif

template <bool BF16, bool PLAN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mla_decode_wide8s_kernel(MlaParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h16 = lane & 15, g = lane >> 4;                  // QK^T / softmax role: head h16 of the wave's 16, key group g
    const int c32 = lane & 31, kg = lane >> 5;                 // P.V role: head c32 of a 32-head block, key half kg
    const int head_blocks = (p.group + 127) / 128;
    int seq, hblk, t_begin, t_end;
    // PLAN: which item of the list this workgroup takes.  The hardware deals workgroup i to XCD i mod 8; inside every run of 16 workgroups,
    // workgroups r and r + 8 -- one XCD -- take the items 2 (r mod 8) and 2 (r mod 8) + 1: the two pieces of a sequence that starts on an even
    // item share an L2 (Q^T crosses from memory once; the rows the pair finish exchanges are read back from that L2 instead of HBM), and
    // any run of items still spreads evenly over the XCDs.  Nothing but traffic depends on the placement being what is assumed here: the
    // pair finish checks it.  (Time: unchanged, 176.6 us either way -- the hand-off is not bound by the fabric; HBM traffic: see DESIGN.)
    const int item_ix = PLAN && ((blockIdx.x | 15u) < gridDim.x) ? (int)((blockIdx.x & ~15u) + ((blockIdx.x & 7u) << 1) + ((blockIdx.x >> 3) & 1u)) : (int)blockIdx.x;
    if constexpr (PLAN) {
        const PlanItem it = plan_item(p.plan, (long long)p.batch * p.kv_heads, item_ix);
        if (it.seq < 0) return;                                // behind the list
        seq = it.seq, hblk = 0, t_begin = it.t_begin, t_end = it.t_end;
    } else {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int u = j / head_blocks;
        hblk = j % head_blocks;
        seq = (u / p.num_splits) * 8 + xcd;                    // (b, kvh) pair
        if (seq >= p.batch * p.kv_heads) return;
        t_begin = t_end = -1;
    }
    const int kvh = seq % p.kv_heads;
    const int b = seq / p.kv_heads;
    if (wave == 0) MLA8S_STAMP(0);
#ifdef MLA8S_STAMPS
    const uint64_t clk0_ = __builtin_amdgcn_s_memtime();
#endif
    const int seq_len = __builtin_amdgcn_readfirstlane(p.seq_lens[b]);
    const int ntiles = (seq_len + kST - 1) / kST;
    if constexpr (PLAN) {
        // the list only decides WHO reads which tiles: pieces are clamped to the sequence's tiles as they are NOW, the last piece runs to
        // their end -- a stale list costs balance, never correctness
        const PlanItem it = plan_item(p.plan, (long long)p.batch * p.kv_heads, item_ix);
        t_begin = min(t_begin, ntiles);
        t_end = it.k == it.n - 1 ? ntiles : min(t_end, ntiles);
    } else {
        const int split = ((blockIdx.x >> 3) / head_blocks) % p.num_splits;
        const int tps = (ntiles + p.num_splits - 1) / p.num_splits;
        t_begin = split * tps;
        t_end = min(ntiles, t_begin + tps);
    }
    const int hg = hblk * 128 + wave * 16 + h16;
    const bool head_ok = hg < p.group;
    const bool wave_active = hblk * 128 + wave * 16 < p.group;       // wave-uniform; idle waves still feed the DMA and own a P.V slice
    const int head = kvh * p.group + hg;
    // DMA destinations (M0) are LDS byte addresses written as literals: the dynamic LDS block of a kernel without static LDS starts at 0
    if (__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds) != 0u) __builtin_trap();
    const CtxS cx{&p, b, seq_len, wave, ntiles, (uint32_t)lane * 16u, __builtin_ctz(p.page_size),
                  p.k_nope + (int64_t)kvh * p.kn_sh, p.k_rope + (int64_t)kvh * p.kr_sh,
                  (uint32_t)p.kn_sblk, (uint32_t)p.kn_srow, (uint32_t)p.kr_sblk, (uint32_t)p.kr_srow};

    // Prologue: tiles t_begin .. t_begin + lead - 1 -> slots 0 .., requested BEFORE the Q^T loads (the fill starts while Q^T is in flight; the
    // block ids are scalar loads, nothing of the fill waits for a vector load)
    int blk_next = 0;                                          // block id of tile t + lead, t = the tile at whose top it is read
    if (t_begin < t_end) {
        int id0 = block_id_request(cx, t_begin), id1 = block_id_request(cx, t_begin + 1), id2 = block_id_request(cx, t_begin + 2);
        blk_next = block_id_request(cx, t_begin + kSLead);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(id0), "+s"(id1), "+s"(id2), "+s"(blk_next)::"memory");
        const int ids[3] = {id0, id1, id2};
#pragma unroll
        for (int d = 0; d < kSLead; ++d) {
            const TileS tl = tile_of(cx, t_begin + d, ids[d]);
#pragma unroll
            for (int i = 0; i < 5; ++i) issue_piece(cx, tl, (uint32_t)(d * kSSlotBytes), i, true);
        }
    }
    // Q^T fragments (B operand of 16x16x32): lane (h16, g) holds q[head][32 ks + 8 g .. +8]
    s16x8 qf[18];
    {
        const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)(head_ok ? head : 0) * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            if (head_ok) qf[ks] = *(const s16x8 *)(qrow + ks * 32 + g * 8);
            else qf[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float nm = 0.f, l_run = 0.f;                               // nm = -(softmax reference) once the first tile has set it
    if (threadIdx.x == 0) *(uint32_t *)(lds + kSFlagOff) = 0;
    const float cs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.sm_scale * 1.4426950408889634f)));
    // vmcnt(0): Q^T resident -- and with it the whole prologue fill (in-order return): no compiler-visible vector load is pending in the
    // loop, and the first two waits at the top of a tile find their pieces landed
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const uint32_t lane16 = cx.lane16;
    // ---- O^T[d, head] += V^T . P^T: wave w owns dims 128 (w >> 1) + {16 dbl + 0..15, 64 + 16 dbl + 0..15} for dbl = 2 (w & 1) + {0, 1};
    // accumulator block dl * 4 + hb = those 32 dims (dbl = 2 (w & 1) + dl) x heads 32 hb .. +31.  Starts behind barrier B.
    const int c16 = lane & 15, q16 = (lane >> 4) & 1;
    const uint32_t v_lane = (uint32_t)((4 * kg + (c16 >> 2)) * kNopeStride + (wave >> 1) * 256 + q16 * 128 + (wave & 1) * 64 + (c16 & 3) * 8);
    // V fragment of step `step` of the tile in slot SLOT (keys 16 kk + {4 kg + 0..3, 8 + 4 kg + 0..3}, step = kk * 2 + dl); fragment 0 is
    // requested by the caller AHEAD of barrier B (the tile has been complete since barrier A: only P^T needs that barrier)
    auto v_frag = [&](auto slot_tag, int step) -> s16x8 {
        constexpr int SLOT = decltype(slot_tag)::value;
        const uint8_t *vlo = lds + SLOT * kSSlotBytes + v_lane;
        const int off = (step >> 1) * 16 * kNopeStride + (step & 1) * 32;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vlo + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vlo + off + 8 * kNopeStride));
        return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto pv = [&](auto slot_tag, s16x8 af0, auto &&under_first_reads) {
        const uint8_t *pb = lds + kPxOff + opaque(lane16);
        auto lda = [&](int step) -> s16x8 { return v_frag(slot_tag, step); };
        auto ldp = [&](int hb, int kk) -> s16x8 { return *(const s16x8 *)(pb + (hb * 2 + kk) * 1024); };
        // 16 MFMAs in the order (kk, dl, hb); operands requested ahead: V fragments one step (4 MFMAs), P fragments three MFMAs
        s16x8 af[2], pfr[4];
        __builtin_amdgcn_sched_barrier(0);
        af[0] = af0;
        pfr[0] = ldp(0, 0);
        pfr[1] = ldp(1, 0);
        pfr[2] = ldp(2, 0);
        __builtin_amdgcn_sched_barrier(0);
        under_first_reads();                                   // VALU work that waits for nothing: runs while the first P^T fragments travel
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int step = i >> 2, hb = i & 3;               // step = kk * 2 + dl
            __builtin_amdgcn_sched_barrier(0);
            if (hb == 0 && step + 1 < 4) af[(step + 1) & 1] = lda(step + 1);
            if (i + 3 < 16) pfr[(i + 3) & 3] = ldp((i + 3) & 3, (i + 3) >> 3);
            __builtin_amdgcn_sched_barrier(0);
            const int a = (step & 1) * 4 + hb;
            acc[a] = mfma32<BF16>(af[step & 1], pfr[i & 3], acc[a]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // Top of tile t (its slot = SLOT): this wave's pieces of t have landed (its 5 (lead - 1) youngest operations -- the pieces of the tiles
    // behind t -- may still be in flight), barrier A: tile t is complete in LDS, everybody is done with tile t - 1 (the pieces of t + lead go
    // to a slot that is free by then) and with the exchange buffer.  The block id of t + lead was read during the P.V phase of tile t - 1.
    auto tile_top = [&](int t) -> TileS {
#ifdef MLA8S_NO_DMA
        __syncthreads();
        return TileS{nullptr, nullptr, 0};
#else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (kSLead - 1)) : "memory");
        __syncthreads();
        return tile_of(cx, t + kSLead, blk_next);
#endif
    };
    // P(t) . V(t) with the block id of tile t + 4 fetched in its shadow
    auto pv_and_next_id = [&](auto slot_tag, int t, s16x8 af0, auto &&under_first_reads) {
#ifndef MLA8S_NO_DMA
        int id = block_id_request(cx, t + 1 + kSLead);
#endif
#ifndef MLA8S_NO_PV
        pv(slot_tag, af0, under_first_reads);
#else
        under_first_reads();
#endif
#ifndef MLA8S_NO_DMA
        block_id_wait(id);
        blk_next = id;
#endif
    };
    const int hbw = wave >> 1;                                  // exchange-buffer coordinates of this wave's P^T pieces: consumer lane
    const int lc = (g & 1) * 32 + (wave & 1) * 16 + h16;        // (kg = g & 1, c32 = 16 (w & 1) + h16), half g >> 1 of its 16 bytes
    const uint32_t pdst_off = (uint32_t)((hbw * 2 * 64 + lc) * 16 + (g >> 1) * 8);      // + kPxOff where it is used
    if (!wave_active) {                                        // no heads of its own: P = 0 for its block, DMA share and P.V slice as usual
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) *(uint2 *)(lds + kPxOff + pdst_off + kb * 1024) = uint2{0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto idle = [&](auto slot_tag, int t) {
            constexpr int SLOT = decltype(slot_tag)::value;
            const TileS tl = tile_top(t);
            constexpr uint32_t nslot = (uint32_t)(((SLOT + kSLead) % kSSlots) * kSSlotBytes);
#pragma unroll
            for (int i = 0; i < 5; ++i) issue_piece(cx, tl, nslot, i);
            const s16x8 af0 = v_frag(slot_tag, 0);
            asm volatile("s_barrier" ::: "memory");              // barrier B
            pv_and_next_id(slot_tag, t, af0, [] {});
        };
        for (int t = t_begin; t < t_end;) {
            idle(SlotTag<0>{}, t);
            if (++t >= t_end) break;
            idle(SlotTag<1>{}, t);
            if (++t >= t_end) break;
            idle(SlotTag<2>{}, t);
            if (++t >= t_end) break;
            idle(SlotTag<3>{}, t);
            ++t;
        }
    }

    // ---- S^T[key, head] = K . Q^T: 18 k-steps of 32 dims x 2 key blocks of 16, K fragments kAhead deep in front of their MFMA, a DMA
    // piece (and the scalar arithmetic of its address) every 7 MFMAs
    const uint32_t a_lane = (uint32_t)(h16 * kNopeStride + g * 16);
    const uint32_t r_lane = (uint32_t)(kST * kNopeStride + h16 * kRopeStride + ((g ^ (h16 & 7)) << 4));
    auto qk = [&](auto slot_tag, const TileS &tl, f32x4 &s0, f32x4 &s1) {
        constexpr int SLOT = decltype(slot_tag)::value;
        constexpr uint32_t nslot = (uint32_t)(((SLOT + kSLead) % kSSlots) * kSSlotBytes);
        const uint8_t *abase = lds + SLOT * kSSlotBytes + a_lane;
        const uint32_t r_off = opaque(r_lane);
        const uint8_t *r0 = lds + (SLOT * kSSlotBytes + r_off);                    // chunk (ks - 16) * 4 + g, swizzled by h16 & 7:
        const uint8_t *r1 = lds + ((SLOT * kSSlotBytes + r_off) ^ 64u);            // the second k-step is chunk ^ 4 (rows are 128-B aligned)
        auto lda = [&](int step) -> s16x8 {                    // step = ks * 2 + kb; key 16 kb + h16, dims 32 ks + 8 g .. +8
            const int ks = step >> 1, kb = step & 1;
            if (ks < 16) return *(const s16x8 *)(abase + kb * 16 * kNopeStride + ks * 64);
            return *(const s16x8 *)((ks == 16 ? r0 : r1) + kb * 16 * kRopeStride);
        };
        constexpr int kAhead = MLA8S_AHEAD, kRing = kAhead + 1;
        s16x8 af[kRing];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pre = 0; pre < kAhead; ++pre) af[pre] = lda(pre);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            __builtin_amdgcn_sched_barrier(0);
#ifdef MLA8S_HALF_QK_READS   // timing probe: every second K fragment is not read (the MFMA takes a stale one; results are garbage) -- what would half the
                             // QK^T operand traffic be worth?  176.3 -> 171.0 us at C4 (merge form), QK^T 1748 -> 1602 clocks per tile: 3 %, the
                             // ceiling of a k-split QK^T on 32x32x16 before its partial-sum exchange and third barrier are paid for.
            if (step + kAhead < 36 && !((step + kAhead) & 1)) af[(step + kAhead) % kRing] = lda(step + kAhead);
#else
            if (step + kAhead < 36) af[(step + kAhead) % kRing] = lda(step + kAhead);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (step == 0) mfma16_first<BF16>(s0, af[0], qf[0]);
            else if (step == 1) mfma16_first<BF16>(s1, af[1 % kRing], qf[0]);
            else if (step & 1) mfma16_acc<BF16>(s1, af[step % kRing], qf[step >> 1]);
            else mfma16_acc<BF16>(s0, af[step % kRing], qf[step >> 1]);
#if MLA8S_DMA_PLACE == 0
            if (step % 7 == 3) issue_piece(cx, tl, nslot, (step / 7 + 4) % 5);          // steps 3, 10, 17, 24, 31 -> pieces 4, 0, 1, 2, 3
#endif
        }
        mfma16_settle(s0, s1);
        __builtin_amdgcn_sched_barrier(0);
    };

#if MLA8S_PRIO == 1
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    constexpr float kGuardP = BF16 ? 0x1p64f : 0x1p11f;         // largest P (relative to the softmax reference) the accumulators are sized for
    if (wave_active) {
#ifdef MLA8S_STAMPS
        uint32_t c0_ = 0;
        if (lane < 8) ((uint32_t *)(lds + kSLds))[wave * 8 + lane] = 0;
        const uint32_t t_loop0_ = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
        auto body = [&](auto slot_tag, int t) {
#ifdef MLA8S_STAMPS
            c0_ = (uint32_t)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (kSLead - 1)) : "memory");
            MLA8S_TICK(0);
#endif
            const TileS tl = tile_top(t);
            MLA8S_TICK(1);
            f32x4 s0, s1;
#ifdef MLA8S_NO_QK           // timing probe: fill + softmax + P.V only
            s0 = s1 = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                constexpr uint32_t nslot = (uint32_t)(((decltype(slot_tag)::value + kSLead) % kSSlots) * kSSlotBytes);
                for (int i = 0; i < 5; ++i) issue_piece(cx, tl, nslot, i);
            }
#else
            qk(slot_tag, tl, s0, s1);
#endif
            MLA8S_TICK(2);
            // lane (h16, g) holds head h16, keys 16 kb + 4 g + i.  Only the tile that crosses seq_len needs the mask.
            if ((t + 1) * kST > seq_len) {
                asm volatile("" ::: "memory");
                const int kbase = t * kST + (int)((opaque(lane16) >> 8) << 2);      // 4 g
                const float ninf = __uint_as_float(opaque(0xff800000u));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (kbase + i >= seq_len) s0[i] = ninf;
                    if (kbase + 16 + i >= seq_len) s1[i] = ninf;
                }
            }
            if (t == t_begin) {                           // the softmax reference of this head: first tile's maximum over all 32 keys
                float tmax = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
                const int self4 = (int)(opaque(lane16) >> 2);  // 4 x lane: byte index of ds_bpermute
                tmax = fmaxf(tmax, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(self4 ^ 64, __builtin_bit_cast(int, tmax))));
                tmax = fmaxf(tmax, __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(self4 ^ 128, __builtin_bit_cast(int, tmax))));
                nm = -(tmax * cs);                             // sm_scale > 0: max commutes with the scaling; a tile below seq_len's end
                                                               // always holds a key, so the reference is finite
            }
            // The critical path of the phase is scores -> exp2 -> P^T in LDS -> barrier B; everything else of the softmax (the running sum, the
            // largest P of the tile for the outgrown-reference check) runs behind the LDS write, the check itself behind barrier B while
            // the first P^T fragments of the P.V phase travel.
            float e[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e[i] = __builtin_amdgcn_exp2f(fma_s(s0[i], cs, nm));
                e[4 + i] = __builtin_amdgcn_exp2f(fma_s(s1[i], cs, nm));
            }
            uint8_t *pdst = lds + kPxOff + pdst_off;
            *(uint2 *)(pdst) = uint2{pack2<BF16>(e[0], e[1]), pack2<BF16>(e[2], e[3])};
            *(uint2 *)(pdst + 1024) = uint2{pack2<BF16>(e[4], e[5]), pack2<BF16>(e[6], e[7])};
            const s16x8 af0 = v_frag(slot_tag, 0);             // first V fragment of the P.V phase: in flight across barrier B
#if MLA8S_DMA_PLACE == 1
            {
                constexpr uint32_t nslot1 = (uint32_t)(((decltype(slot_tag)::value + kSLead) % kSSlots) * kSSlotBytes);
#pragma unroll
                for (int i = 0; i < 5; ++i) issue_piece(cx, tl, nslot1, (i + 4) % 5);
            }
#endif
            l_run += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
            // P >= 0: float order = order of the bit patterns (v_max3_u32, no NaN canonicalisation in front of every operand)
            const uint32_t emax = max3u(max3u(__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2])),
                                        max3u(__float_as_uint(e[3]), __float_as_uint(e[4]), __float_as_uint(e[5])),
                                        max(__float_as_uint(e[6]), __float_as_uint(e[7])));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MLA8S_TICK(3);
            asm volatile("s_barrier" ::: "memory");               // barrier B: P^T(t) complete
            MLA8S_TICK(4);
            pv_and_next_id(slot_tag, t, af0, [&] {
#if MLA8S_DMA_PLACE == 2
                {
                    constexpr uint32_t nslot2 = (uint32_t)(((decltype(slot_tag)::value + kSLead) % kSSlots) * kSSlotBytes);
#pragma unroll
                    for (int i = 0; i < 5; ++i) issue_piece(cx, tl, nslot2, (i + 4) % 5);
                }
#endif
                // a P above 2^kGuard (the first tile's are <= 1): the reference has been outgrown, the sequence is recomputed exactly
                if (__any(emax > __float_as_uint(kGuardP))) {
#if !defined(MLA8S_NO_DMA) && !defined(MLA8S_NO_QK) && !defined(MLA8S_NO_PV)
                    *(uint32_t *)(lds + kSFlagOff) = opaque(1u);
#endif
                }
            });
            MLA8S_TICK(5);
        };
        for (int t = t_begin; t < t_end;) {
            body(SlotTag<0>{}, t);
            if (++t >= t_end) break;
            body(SlotTag<1>{}, t);
            if (++t >= t_end) break;
            body(SlotTag<2>{}, t);
            if (++t >= t_end) break;
            body(SlotTag<3>{}, t);
            ++t;
        }
#ifdef MLA8S_STAMPS
        if (lane == 0 && blockIdx.x < 256) {
            for (int i = 0; i < 6; ++i) g_mla8s_phase[blockIdx.x][wave][i] = (float)((volatile uint32_t *)(lds + kSLds))[wave * 8 + i] / (float)max(1, t_end - t_begin);
            g_mla8s_phase[blockIdx.x][wave][6] = (float)((uint32_t)__builtin_amdgcn_s_memtime() - t_loop0_);
            g_mla8s_phase[blockIdx.x][wave][7] = (float)(t_end - t_begin);
        }
#endif
        l_run += __shfl_xor(l_run, 16, 64);                    // the four key groups of a head
        l_run += __shfl_xor(l_run, 32, 64);
    }
    const float m_run = (wave_active && t_begin < t_end) ? -nm : -INFINITY;
    if (wave == 0) MLA8S_STAMP(1);
#ifdef MLA8S_STAMPS
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) g_mla8s_stamp[blockIdx.x][7] = __builtin_amdgcn_s_memtime() - clk0_;      // shader clocks, start -> loop end
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // fills issued past the last tile
    __syncthreads();
    const bool flagged_local = *(volatile uint32_t *)(lds + kSFlagOff) != 0;
    if (threadIdx.x == 0 && flagged_local) p.fix_flags[b * p.kv_heads + kvh] = p.fix_epoch;

    // ---- epilogue: acc[dl * 4 + hb][4 rg + i] = O^T[d][head 32 hb + c32], d = 128 (w >> 1) + 64 (rg >> 1) + 16 (2 (w & 1) + dl) + 8 (rg & 1) + 4 kg + i;
    // the softmax statistics of a head live in the wave that owns it and reach the others through LDS
    float *lmb = (float *)(lds + kPxOff);                      // [0..127] l, [128..255] m (all P.V reads are behind the barrier above)
    if (g == 0) {
        lmb[wave * 16 + h16] = wave_active ? l_run : 0.f;
        lmb[128 + wave * 16 + h16] = m_run;
    }
    __syncthreads();
    int nsplits, pmul, piece;
    bool near = true;                                          // (uniform form: the splits of a sequence are dealt to one XCD)
    int64_t pbase, pbase_partner;
    if constexpr (PLAN) {
        const PlanItem it = plan_item(p.plan, (long long)p.batch * p.kv_heads, item_ix);
        nsplits = it.n, piece = it.k;
        pmul = 1, pbase = ((int64_t)item_ix - kvh) * p.group;
        pbase_partner = pbase + (piece == 0 ? p.group : -p.group);       // the pieces of a sequence are consecutive items
        // partner expected on this XCD: the pair starts on an even item of a full run of 16 workgroups
        near = ((item_ix - piece) & 1) == 0 && ((blockIdx.x | 15u) < gridDim.x);
    } else {
        nsplits = p.num_splits, piece = ((blockIdx.x >> 3) / head_blocks) % p.num_splits;
        pmul = p.num_splits, pbase = (int64_t)b * p.q_heads * p.num_splits + piece;
        pbase_partner = pbase + (piece == 0 ? 1 : -1);
    }
    auto pslot = [&](int headx) -> int64_t { return pbase + (int64_t)headx * pmul; };
    const bool finals = nsplits == 1;                          // this workgroup writes output rows itself (no merge launch)
    // a sequence in TWO pieces: each workgroup publishes the partial rows of the OTHER piece's heads, then finishes its own heads (piece 0:
    // head blocks 0 and 1, piece 1: blocks 2 and 3) from its accumulators and the partner's rows (below); the merge launch finds nothing
    // to do for it
    const bool pair = p.pair_flags != nullptr && nsplits == 2 && head_blocks == 1;
    if (finals && flagged_local) {                             // outgrown softmax reference: exact slow path, one head per wave at a time
        for (int i = 0; i < 16; ++i) {
            const int hg2 = hblk * 128 + wave * 16 + i;
            if (hg2 < p.group) mla_recompute_head<BF16>(p, b, kvh * p.group + hg2, lane);
        }
        return;
    }
    // Rows leave through a wave-private LDS tile (the KV ring is free now), one head block at a time: [32 heads][64 dims] fp32, a tile
    // row = this wave's dims in the order (64-dim half, dl, 16): every 8 lanes then store 128 contiguous bytes of one head.
    constexpr int kEpiRow = 64 * 4 + 16;
    uint8_t *const tile = lds + wave * (32 * kEpiRow);
    const int dcol0 = (wave >> 1) * 128 + (wave & 1) * 32;      // global dim of tile column 0; columns 32.. are 64 dims further
    // o8[it] = four consecutive dims (tile chunk lane & 15) of head 4 it + (lane >> 4) of head block hb
    auto tile_write = [&](auto hb_tag) {
        constexpr int hb = decltype(hb_tag)::value;            // static accumulator indices
#pragma unroll
        for (int dl = 0; dl < 2; ++dl)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int col = (rg >> 1) * 32 + dl * 16 + (rg & 1) * 8 + 4 * kg;
                const f32x16 &a = acc[dl * 4 + hb];
                *(f32x4 *)(tile + c32 * kEpiRow + col * 4) = f32x4{a[4 * rg + 0], a[4 * rg + 1], a[4 * rg + 2], a[4 * rg + 3]};
            }
    };
    auto transpose_block = [&](auto hb_tag, f32x4 (&o8)[8]) {
        tile_write(hb_tag);
        // wave-private tile: LDS operations of one wave complete in order, no barrier
#pragma unroll
        for (int it = 0; it < 8; ++it) o8[it] = *(const f32x4 *)(tile + (it * 4 + (lane >> 4)) * kEpiRow + (lane & 15) * 16);
    };
    const int ch = lane & 15;
    const int dlane = dcol0 + (ch >> 3) * 64 + (ch & 7) * 4;     // this lane's four dims of a row
    // kind 0: output rows (this workgroup holds the whole sum), 1: partial rows, 2: partial rows written through (the pair's export)
    auto rows_out = [&](auto hb_tag, int kind) {
        constexpr int hb = decltype(hb_tag)::value;
        if (hblk * 128 + hb * 32 >= p.group) return;
        f32x4 o8[8];
        transpose_block(hb_tag, o8);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int hl = it * 4 + (lane >> 4);
            const int hgx = hblk * 128 + hb * 32 + hl;
            const int headx = kvh * p.group + min(hgx, p.group - 1);
            if (hgx >= p.group) continue;
            if (kind == 0) {
                const float l_h = lmb[hb * 32 + hl];
                const float inv = l_h > 0.f ? 1.f / l_h : 0.f;
                const uint32_t w0 = (uint32_t)cvt_out<BF16>(o8[it][0] * inv) | ((uint32_t)cvt_out<BF16>(o8[it][1] * inv) << 16);
                const uint32_t w1 = (uint32_t)cvt_out<BF16>(o8[it][2] * inv) | ((uint32_t)cvt_out<BF16>(o8[it][3] * inv) << 16);
                *(uint2 *)(p.out + (int64_t)b * p.o_sb + (int64_t)headx * p.o_sh + dlane) = uint2{w0, w1};
            } else if (kind == 2) {
                st_sc1_x4(p.ws_o + pslot(headx) * kDN + dlane, o8[it]);      // visible to the partner behind a vmcnt drain, no release fence
            } else {
                *(f32x4 *)(p.ws_o + pslot(headx) * kDN + dlane) = o8[it];
            }
        }
    };
    // softmax statistics of the workgroup's heads: [m, l] per partial row (one lane per head)
    if (!finals && wave == 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int hgx = hblk * 128 + r * 64 + lane;
            if (hgx < p.group) {
                const int64_t idx = pslot(kvh * p.group + hgx);
                st_agent_f32(p.ws_ml + idx * 2 + 0, lmb[128 + r * 64 + lane]);
                st_agent_f32(p.ws_ml + idx * 2 + 1, lmb[r * 64 + lane]);
            }
        }
    }
    auto leave_to_merge_kernel = [&]() {
        if (threadIdx.x == 0 && p.need_merge) __hip_atomic_store(p.need_merge, p.pair_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (!pair) {
        if (!finals) leave_to_merge_kernel();
        const int kind = finals ? 0 : 1;
        rows_out(SlotTag<0>{}, kind);
        rows_out(SlotTag<1>{}, kind);
        rows_out(SlotTag<2>{}, kind);
        rows_out(SlotTag<3>{}, kind);
        if (wave == 0) MLA8S_STAMP(2);
        return;
    }

    // ---- the pair.  Export (all eight waves: every wave holds 64 dims of every head): the rows of the partner's heads, written through.
    // "My rows are visible" -> the partner's word (bounded wait) -> own heads = w0 a0 + w1 a1 in piece order, exactly the merge kernel's
    // sums, from the own accumulators and the partner's rows (write-through stores are read with sc1 loads: no acquire).  A partner that
    // does not show up in time (not resident: more items than the chip runs at once) costs nothing but the wait: this workgroup then
    // writes the rows of its own heads as well and leaves them unmarked -- both partials of those heads are in the workspace and the merge
    // kernel, which skips only heads whose piece carries the mark, does the work.
    // Partner expected on this XCD (`near`): plain stores -- the rows wait in the shared L2 for the partner's sc1 (L1-bypassing) loads and do
    // not have to come back from HBM; otherwise write-through.  Both workgroups publish their XCC id with the meeting word: a pair that
    // expected to be neighbours and is not (placement is the hardware's business) takes the bounded wait's way out -- the merge kernel, for
    // which the kernel boundary makes every row visible.
    const int xkind = near ? 1 : 2;
    if (piece == 0) rows_out(SlotTag<2>{}, xkind), rows_out(SlotTag<3>{}, xkind);
    else rows_out(SlotTag<0>{}, xkind), rows_out(SlotTag<1>{}, xkind);
    if (wave == 0) MLA8S_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave: its write-through stores have left
    __syncthreads();
    if (wave == 0) MLA8S_STAMP(3);
    uint32_t *const ok_word = (uint32_t *)(lmb + 256);
    const int64_t fbase = ((int64_t)b * p.kv_heads + kvh) * 2;
    if (threadIdx.x == 0) {
        const uint64_t xcc = (uint64_t)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);      // HW_REG_XCC_ID[3:0]
        const uint64_t tag = p.pair_tag & ~7ull;
        if (!(p.pair_withhold && piece == 1))
            __hip_atomic_store(p.pair_flags + fbase + piece, tag | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t ok = 1;
        uint64_t seen;
        while (((seen = __hip_atomic_load(p.pair_flags + fbase + (piece ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & ~7ull) != tag) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000ull) {      // 200 us at 100 MHz
                ok = 0;
                break;
            }
        }
        if (ok && near && (seen & 7ull) != xcc) ok = 0;       // expected neighbours on different XCDs: plain rows may not be visible here
        *ok_word = ok;
    }
    __syncthreads();
    if (wave == 0) MLA8S_STAMP(4);
    if (flagged_local) {                                       // (the merge kernel recomputes the sequence)
        leave_to_merge_kernel();
        return;
    }
    if (*(volatile uint32_t *)ok_word == 0) {
        leave_to_merge_kernel();
        if (piece == 0) rows_out(SlotTag<0>{}, 1), rows_out(SlotTag<1>{}, 1);
        else rows_out(SlotTag<2>{}, 1), rows_out(SlotTag<3>{}, 1);
        return;
    }
    // "this piece finishes its heads": the sign of their sums (nobody else reads those words before the merge kernel)
    if (wave == 0) {
        const int hgx = piece * 64 + lane;
        if (hgx < p.group) st_agent_f32(p.ws_ml + pslot(kvh * p.group + hgx) * 2 + 1, -lmb[hgx]);
    }
    // The partner's rows of this workgroup's heads: 16 write-through-coherent loads per lane, all in flight before the first use.  Lane
    // mapping of the finish: head 8 it + (lane >> 3) of a head block, EIGHT consecutive dims (lane & 7) of the wave's 64 -- one 16-byte
    // output store per lane and head (the four-dim mapping of the export needs twice as many, and the store tail is issue-bound).
    const int ch8 = lane & 7;
    const int dlane8 = dcol0 + (ch8 >> 2) * 64 + (ch8 & 3) * 8;
    const float *const prow = p.ws_o + pbase_partner * kDN;    // + head * pmul * kDN + dim
    f32x4 pr[2][8];                                            // [head block of the pair][2 it + half]
    auto partner_rows = [&](int hb, f32x4 (&r)[8]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int hgx = min(hb * 32 + it * 8 + (lane >> 3), p.group - 1);
            const float *src = prow + (int64_t)(kvh * p.group + hgx) * pmul * kDN + dlane8;
            r[2 * it] = ld_sc1_x4(src);
            r[2 * it + 1] = ld_sc1_x4(src + 4);
        }
    };
    partner_rows(piece * 2, pr[0]);
    partner_rows(piece * 2 + 1, pr[1]);
    // the weights of the workgroup's 64 heads, once per head (lane = head) instead of once per (head, dims) item: both pieces' statistics ->
    // [w0, w1, 1 / L] in piece order, exactly the merge kernel's arithmetic, through a wave-private LDS table
    float *const wts = (float *)(lds + 8 * (32 * kEpiRow)) + wave * 256;
    {
        const int hgx = min(piece * 64 + lane, p.group - 1);
        const int64_t idx = pbase_partner + (int64_t)(kvh * p.group + hgx) * pmul;
        const float m_par = __hip_atomic_load(p.ws_ml + idx * 2 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float l_par = __hip_atomic_load(p.ws_ml + idx * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float m_own = lmb[128 + hgx], l_own = lmb[hgx];
        const float m0 = piece == 0 ? m_own : m_par, m1 = piece == 0 ? m_par : m_own;
        const float l0 = piece == 0 ? l_own : l_par, l1 = piece == 0 ? l_par : l_own;
        const float M = fmaxf(m0, m1);
        const float w0 = m0 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m0 - M);
        const float w1 = m1 == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m1 - M);
        float L = 0.f;
        if (w0 != 0.f) L += w0 * l0;
        if (w1 != 0.f) L += w1 * l1;
        *(f32x4 *)(wts + lane * 4) = f32x4{w0, w1, L > 0.f ? 1.f / L : 0.f, 0.f};
    }
    ld_sc1_wait(pr[0]);
    ld_sc1_wait(pr[1]);
    auto finish = [&](auto hb_tag, const f32x4 (&r)[8]) {
        constexpr int hb = decltype(hb_tag)::value;
        if (hb * 32 >= p.group) return;
        tile_write(hb_tag);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int hl = it * 8 + (lane >> 3);
            const int hgx = hb * 32 + hl;
            const f32x4 own_lo = *(const f32x4 *)(tile + hl * kEpiRow + ch8 * 32), own_hi = *(const f32x4 *)(tile + hl * kEpiRow + ch8 * 32 + 16);
            const f32x4 wt = *(const f32x4 *)(wts + (min(hgx, p.group - 1) - piece * 64) * 4);
            if (hgx >= p.group) continue;
            const int headx = kvh * p.group + hgx;
            const float w0 = wt[0], w1 = wt[1], inv = wt[2];
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                      // w0 a0 + w1 a1 in piece order, a term of weight 0 skipped (the merge kernel's sums)
                const float a0l = piece == 0 ? own_lo[j] : r[2 * it][j], a0h = piece == 0 ? own_hi[j] : r[2 * it + 1][j];
                const float a1l = piece == 0 ? r[2 * it][j] : own_lo[j], a1h = piece == 0 ? r[2 * it + 1][j] : own_hi[j];
                if (w0 != 0.f) o[j] += w0 * a0l, o[4 + j] += w0 * a0h;
                if (w1 != 0.f) o[j] += w1 * a1l, o[4 + j] += w1 * a1h;
            }
            u32x4 x;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = (uint32_t)cvt_out<BF16>(o[2 * j] * inv) | ((uint32_t)cvt_out<BF16>(o[2 * j + 1] * inv) << 16);
            *(u32x4 *)(p.out + (int64_t)b * p.o_sb + (int64_t)headx * p.o_sh + dlane8) = x;
        }
    };
    if (piece == 0) finish(SlotTag<0>{}, pr[0]), finish(SlotTag<1>{}, pr[1]);
    else finish(SlotTag<2>{}, pr[0]), finish(SlotTag<3>{}, pr[1]);
#ifdef MLA8S_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0) MLA8S_STAMP(6);
#endif
}

}  // namespace

#ifdef MLA8S_STAMPS
extern "C" int mi_mla8s_phases(void *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mla8s_phase), sizeof(float) * 256 * 8 * 8); }
extern "C" int mi_mla8s_stamps(void *host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_mla8s_stamp), sizeof(unsigned long long) * 1024 * 8); }
#endif

#ifdef MLA8S_STAMPS
#define MLA8S_LDS (kSLds + 256)
#else
#define MLA8S_LDS kSLds
#endif
void launch_mla_wide8s(const MlaParams &p, int dtype, long long units, hipStream_t st)
{
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8s_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLA8S_LDS);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8s_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLA8S_LDS);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8s_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLA8S_LDS);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8s_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLA8S_LDS);
    }
    const int head_blocks = (p.group + 127) / 128;
    const long long seqs = p.plan ? 0 : units / p.num_splits;  // (sequence, kv head) pairs, 8 per grid row of XCDs
    dim3 grid(p.plan ? (unsigned)units : (unsigned)(((seqs + 7) / 8) * 8 * p.num_splits * head_blocks));
    if (p.plan) {
        if (dtype == MI_DTYPE_BF16) mla_decode_wide8s_kernel<true, true><<<grid, 512, MLA8S_LDS, st>>>(p);
        else mla_decode_wide8s_kernel<false, true><<<grid, 512, MLA8S_LDS, st>>>(p);
    } else {
        if (dtype == MI_DTYPE_BF16) mla_decode_wide8s_kernel<true, false><<<grid, 512, MLA8S_LDS, st>>>(p);
        else mla_decode_wide8s_kernel<false, false><<<grid, 512, MLA8S_LDS, st>>>(p);
    }
}

}  // namespace mi_sgl
