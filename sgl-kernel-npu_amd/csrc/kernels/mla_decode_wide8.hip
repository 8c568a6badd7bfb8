// Paged MLA decode, wide variant on EIGHT waves (two per SIMD) -- kv groups of more than 64 heads, BASELINE C4.
// Reference replaced: python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:5-230 (same numerics contract as
// mla_decode_wide.hip, whose 4-wave form this kernel supersedes: there every DMA issue, LDS operand wait and barrier of a wave
// stopped its SIMD's matrix pipe -- one wave per SIMD has nobody to cover for it -- and MFMA issue was 53 % of the tile time).
//
// One workgroup = ALL 128 heads of a (sequence, KV split), 512 threads.  Register budget per wave: 256 (two waves share a SIMD's
// 512): 128 accumulator registers + Q^T of 16 heads (72) + operands.
//   QK^T + softmax by HEAD:  wave w owns heads 16 w .. 16 w + 15; S^T[32 keys, 16 heads] = K . Q^T as two 16-key blocks on
//     v_mfma_f32_16x16x32 (A = K rows from LDS, ds_read_b128; B = Q^T resident in registers), two independent accumulator chains.
//   P . V by OUTPUT DIMENSION: wave w owns 64 of the 512 dims for all 128 heads (8 accumulator blocks of 32 dims x 32 heads on
//     v_mfma_f32_32x32x16): a V tile is read from LDS once per workgroup (ds_read_b64_tr_b16), P^T crosses from the head owners to
//     the dimension owners through an 8 KB LDS exchange buffer laid out as the 32x32x16 B operand.
// Tile loop (32 keys), every wave the same:  wait own DMA pieces | barrier A | QK^T(t) with the DMA pieces of tile t+2 between the
// MFMAs | softmax, P^T(t) -> exchange buffer | barrier B | P(t) . V(t).  No software pipelining across the barriers: with two
// waves per SIMD the partner's MFMAs cover a wave's DMA issue, operand waits and softmax.
// Softmax reference: the first tile's maximum per head, never rescaled (see mla_decode_wide.hip); a sequence whose later scores
// outgrow it is flagged and recomputed exactly by the merge kernel's slow path (mla_decode.hip: mla_recompute_head).
// Measured at C4: per tile and wave 5980 cycles -- QK^T 2950 (its 8 x 36 KB of LDS operand reads per tile, one 1 KB
// fragment per 16-cycle MFMA, are the LDS peak rate exactly), barrier A + piece addresses 1090, softmax + publish 490, barrier B 550,
// P.V 900; the own DMA wait is 8 cycles (memory is never late).  The kernel ties with the four-wave one (181 vs 183 us).  A
// K-SPLIT variant (pairs of waves share 32 heads and split the 576 dims on 32x32x16 -- half the LDS traffic -- and exchange half of
// their partial S^T through LDS; three KV slots) was built and is correct, but 128 accumulators + 72 of Q^T + the 16-register S^T
// chain leave no room in 256 registers: five Q^T fragments went to scratch, each reload waits vmcnt(0) = every DMA piece in
// flight, 264 us.  Eight waves double the per-wave overhead registers over the same 512 KB register file; the 4-wave form fits.
// LDS: ring of 4 KV slots (32 keys x (1056 + 128) B), per-wave block-table rings, exchange buffer = 160 KiB exactly; the two
// control words live in the pad bytes of the last K row.  K rows are NOT chunk-swapped here (the 16-row operand fetch of the
// 16x16x32 form is conflict-free under the plain 1056-byte stride, and so is the transposed V fetch).
#include "device_once.h"
#include "mi_sgl_kernels.h"
#include "mla_common.h"

namespace mi_sgl {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int k8T = kWideTile, k8Slots = 4, k8Waves = 8, k8Lead = 2;
constexpr int k8SlotBytes = k8T * kNopeStride + k8T * kRopeStride;          // 37888
constexpr int k8RingEntries = 32;
constexpr int k8RingBytes = k8Waves * k8Slots * k8RingEntries * 4;          // 4096
constexpr int k8POff = k8Slots * k8SlotBytes + k8RingBytes;                 // P^T exchange buffer [head block 4][k-step 2][lane 64] x 16 B
constexpr int k8PBytes = 8192;
constexpr int k8Lds = k8POff + k8PBytes;                                    // 163840 = 160 KiB
constexpr int k8FlagOff = (k8Slots - 1) * k8SlotBytes + (k8T - 1) * kNopeStride + kDN * 2;      // pad of the last K row: never a DMA target
constexpr int k8OpsPerTile = 6;                                             // block ids + 4 K rows + half a rope piece, every wave
static_assert(k8Lds <= 160 * 1024, "LDS budget");

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

// LDS-DMA through inline asm (the compiler then tracks no vector-memory operation in the tile loop and places no vmcnt wait of
// its own; ordering is the explicit s_waitcnt at the top of a tile).  M0 = wave-uniform LDS destination.
__device__ __forceinline__ uint32_t lds_addr8(const void *generic)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)generic;
}
__device__ __forceinline__ void dma16_sbase8(uint32_t dst, const void *sbase, uint32_t voff)     // lane l: 16 B from sbase + voff -> dst + 16 l
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory", "m0");
}
__device__ __forceinline__ void dma16_vaddr8(uint32_t dst, const void *vaddr)                    // active lane l: 16 B from its own address
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(vaddr) : "memory", "m0");
}
__device__ __forceinline__ void dma4_vaddr8(uint32_t dst, const void *vaddr)                     // active lane l: 4 B -> dst + 4 l
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(vaddr) : "memory", "m0");
}

struct Ctx8 {
    const MlaParams *p;
    int b, seq_len, wave, lane;
    uint32_t *ring;                    // this wave's [k8Slots][32] block ids
    uint32_t lds_base, ring_addr;
    int page_shift;
    const uint16_t *kn_base, *kr_base;
    uint32_t kn_sblk, kn_srow, kr_sblk, kr_srow;
};

template <bool POW2> __device__ __forceinline__ int page_of(const Ctx8 &c, int n) { return (POW2 || c.page_shift >= 0) ? n >> c.page_shift : n / c.p->page_size; }
__device__ __forceinline__ int key_of(const Ctx8 &c, int tile)                                   // lanes 32..63 mirror 0..31
{
    int n = tile * k8T + (c.lane & 31);
    n = n < c.seq_len ? n : c.seq_len - 1;
    return n < 0 ? 0 : n;
}
template <bool POW2> __device__ __forceinline__ void issue_rows8(const Ctx8 &c, int tile)                            // block-table entries of `tile` -> ring (1 op)
{
    const int page = page_of<POW2>(c, key_of(c, tile));
    const int32_t *src = c.p->block_table + (int64_t)c.b * c.p->bt_stride + page;
    if (c.lane < k8RingEntries) dma4_vaddr8(c.ring_addr + (uint32_t)((tile & (k8Slots - 1)) * k8RingEntries * 4), src);
}
template <bool POW2> __device__ __forceinline__ TileRows rows8(const Ctx8 &c, int tile)
{
    const int n = key_of(c, tile);
    const int page = page_of<POW2>(c, n);
    const uint32_t row = (POW2 || c.page_shift >= 0) ? (uint32_t)n & (uint32_t)(c.p->page_size - 1) : (uint32_t)(n - page * c.p->page_size);
    const uint32_t blk = c.ring[(tile & (k8Slots - 1)) * k8RingEntries + (c.lane & (k8RingEntries - 1))];
    TileRows r;                        // 32-bit pieces: the launcher sends caches whose strides do not fit to the 64-head kernel
    r.nope = (int64_t)((uint64_t)blk * c.kn_sblk + __umul24(row, c.kn_srow));
    r.rope = (int64_t)((uint64_t)blk * c.kr_sblk + __umul24(row, c.kr_srow));
    return r;
}
// Source addresses of this wave's five DMA pieces of a tile, taken out of the per-lane row offsets right away: the four K-row addresses
// are wave-uniform (SGPR pairs), only the rope piece keeps a per-lane pointer -- the row offsets themselves (4 VGPRs) do not stay live
// through QK^T (the register budget is 256 with 200 of them spoken for).
struct Pieces8 {
    const uint16_t *nope[4];           // K rows wave + 8 idx
    const uint16_t *rope;              // lanes 0..31: 16 B of rope row 4 wave + (lane >> 3)
};
__device__ __forceinline__ Pieces8 pieces8(const Ctx8 &c, const TileRows &rows)
{
    Pieces8 q;
#pragma unroll
    for (int idx = 0; idx < 4; ++idx) {
        const int i = c.wave + 8 * idx;
        const int lo = __builtin_amdgcn_readlane((int)(rows.nope & 0xFFFFFFFFll), i);
        const int hi = __builtin_amdgcn_readlane((int)(rows.nope >> 32), i);
        q.nope[idx] = c.kn_base + (((int64_t)hi << 32) | (uint32_t)lo);
    }
    const int key = c.wave * 4 + ((c.lane >> 3) & 3);
    const int chunk = (c.lane & 7) ^ (key & 7);
    q.rope = c.kr_base + lane_i64(rows.rope, key) + chunk * 8;
    return q;
}
// piece idx 0..3: K row wave + 8 idx (1 KiB); idx 4: rope rows 4 wave .. +3 (4 x 128 B, lanes 0..31).  `slot` = LDS byte address
__device__ __forceinline__ void issue_piece8(const Ctx8 &c, const Pieces8 &q, uint32_t slot, int idx)
{
    if (idx < 4) {
        dma16_sbase8(slot + (uint32_t)((c.wave + 8 * idx) * kNopeStride), q.nope[idx], (uint32_t)(c.lane * 16));
    } else {
        if (c.lane < 32) dma16_vaddr8(slot + (uint32_t)(k8T * kNopeStride + c.wave * 4 * kRopeStride), q.rope);
    }
}

template <bool BF16, bool PLAN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mla_decode_wide8_kernel(MlaParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
#ifdef MLA8_TIMING
    const uint64_t r_start = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h16 = lane & 15, g = lane >> 4;                  // QK^T / softmax role: head h16 of the wave's 16, key group g
    const int c32 = lane & 31, kg = lane >> 5;                 // P.V role: head c32 of a 32-head block, key half kg
    const int head_blocks = (p.group + 127) / 128;
    // workgroups b % 8 run on XCD b % 8: the num_splits workgroups of a (sequence, kv head) share an XCD (their partials meet in its L2)
    // PLAN (template): workgroup = item of the device-built work list (decode_plan.h; the pieces of a sequence are consecutive items).
    // A template parameter, not a run-time branch, and power-of-two page sizes only: the kernel sits at the edge of the register
    // file.  The uniform form re-derives its tile range from kernel arguments; this one LOADS it, so those scalars stay live across the
    // tile loop -- together with the scalars of the any-page-size division path they no longer fit, the overflow goes to a vector register
    // and four Q^T fragments to scratch (a scratch reload waits vmcnt(0), i.e. for every DMA piece in flight).  Without the division
    // path it allocates like the uniform form.  (Other page sizes take the uniform form.)
    int seq, hblk, t_begin, t_end;
    if constexpr (PLAN) {
        // (no separate check against the list's length: every slot behind it is padding, seq = -1)
        const PlanItem it = plan_item(p.plan, (long long)p.batch * p.kv_heads, (int)blockIdx.x);
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
    const int seq_len = p.seq_lens[b];
    if constexpr (PLAN) {
        // the list only decides WHO reads which tiles: the pieces are clamped to the sequence's tiles as they are NOW and the last piece
        // runs to their end, so a list built from older lengths (a step ago, or another batch in the same buffer) costs balance, never
        // correctness
        const int ntiles = (seq_len + k8T - 1) / k8T;
        const PlanItem it = plan_item(p.plan, (long long)p.batch * p.kv_heads, (int)blockIdx.x);
        t_begin = min(t_begin, ntiles);
        t_end = it.k == it.n - 1 ? ntiles : min(t_end, ntiles);
    }
    if constexpr (!PLAN) {
        const int split = ((blockIdx.x >> 3) / head_blocks) % p.num_splits;
        const int ntiles = (seq_len + k8T - 1) / k8T;
        const int tps = (ntiles + p.num_splits - 1) / p.num_splits;
        t_begin = split * tps;
        t_end = min(ntiles, t_begin + tps);
    }
    const int hg = hblk * 128 + wave * 16 + h16;
    const bool head_ok = hg < p.group;
    const bool wave_active = hblk * 128 + wave * 16 < p.group;       // wave-uniform; idle waves still feed the DMA and own a P.V slice
    const int head = kvh * p.group + hg;
    Ctx8 cx{&p, b, seq_len, wave, lane, (uint32_t *)(lds + k8Slots * k8SlotBytes) + wave * k8Slots * k8RingEntries, 0, 0,
            (p.page_size & (p.page_size - 1)) == 0 ? __builtin_ctz(p.page_size) : -1,
            p.k_nope + (int64_t)kvh * p.kn_sh, p.k_rope + (int64_t)kvh * p.kr_sh,
            (uint32_t)p.kn_sblk, (uint32_t)p.kn_srow, (uint32_t)p.kr_sblk, (uint32_t)p.kr_srow};
    cx.lds_base = __builtin_amdgcn_readfirstlane(lds_addr8(lds));
    cx.ring_addr = cx.lds_base + (uint32_t)(k8Slots * k8SlotBytes + wave * k8Slots * k8RingEntries * 4);

    // block ids of the first tiles are requested BEFORE the Q^T loads (see mla_decode_wide.hip): the KV fill then starts while Q^T
    // is still in flight
#pragma unroll
    for (int d = 0; d < k8Lead; ++d) issue_rows8<PLAN>(cx, t_begin + d);
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
    float m_run = -INFINITY, l_run = 0.f;
    if (threadIdx.x == 0) *(uint32_t *)(lds + k8FlagOff) = 0;
    const float cs = p.sm_scale * 1.4426950408889634f;

    // Issue order per tile x and wave: R(x + 4) (block ids), D(x + 2) (5 pieces) = 6 vector-memory operations; the wait at the top
    // of tile x leaves the youngest 6 in flight: this wave's pieces of tile x and the block ids of tile x + 2 have landed.  After
    // barrier A tile x is complete in LDS, everybody is done with tile x - 1 and with the exchange buffer.
    // (Forming the piece addresses of tile x + 2 one iteration earlier, behind the P.V MFMAs of tile x - 1, takes their ~400-cycle
    //  serial chain -- ring read, two 64-bit multiply-adds, eight v_readlane, two ds_bpermute -- off the path between barrier A and the
    //  first QK^T MFMA, but the two registers that then live across the barrier pushed two Q^T fragments into scratch, and a scratch
    //  reload waits vmcnt(0), i.e. for every DMA piece in flight: built, measured slower, removed.)
    auto tile_top = [&](int t) -> Pieces8 {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
        issue_rows8<PLAN>(cx, t + k8Lead + 2);
        return pieces8(cx, rows8<PLAN>(cx, t + k8Lead));
    };
    if (t_begin < t_end) {                                      // prologue in steady-state issue order
        if (__any(head_ok)) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");      // block ids landed; at most the 18 Q^T loads outstanding
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < k8Lead; ++d) {
            issue_rows8<PLAN>(cx, t_begin + d + 2);
            const Pieces8 r = pieces8(cx, rows8<PLAN>(cx, t_begin + d));
#pragma unroll
            for (int i = 0; i < 5; ++i) issue_piece8(cx, r, cx.lds_base + (uint32_t)(((t_begin + d) & (k8Slots - 1)) * k8SlotBytes), i);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0): Q^T resident, no compiler-visible vector load pending in the loop

    uint8_t *const pbuf = lds + k8POff;
    // ---- O^T[d, head] += V^T . P^T: wave w owns dims 128 (w >> 1) + {16 dbl + 0..15, 64 + 16 dbl + 0..15} for dbl = 2 (w & 1) + {0, 1};
    // accumulator block dl * 4 + hb = those 32 dims (dbl = 2 (w & 1) + dl) x heads 32 hb .. +31.  Starts behind barrier B.
    auto pv = [&](int t) {
        const uint8_t *buf = lds + (t & (k8Slots - 1)) * k8SlotBytes;
        const uint8_t *pb = pbuf + lane * 16;
        const int c16 = lane & 15, q16 = (lane >> 4) & 1;
        const uint8_t *vlo = buf + (4 * kg + (c16 >> 2)) * kNopeStride + (wave >> 1) * 256 + q16 * 128 + (wave & 1) * 64 + (c16 & 3) * 8;
        const uint8_t *vhi = vlo + 8 * kNopeStride;
        auto lda = [&](int step) -> s16x8 {                    // step = kk * 2 + dl: keys 16 kk + {4 kg + 0..3, 8 + 4 kg + 0..3}
            const int off = (step >> 1) * 16 * kNopeStride + (step & 1) * 32;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vlo + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(vhi + off));
            return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto ldp = [&](int hb, int kk) -> s16x8 { return *(const s16x8 *)(pb + (hb * 2 + kk) * 1024); };
        // 16 MFMAs in the order (kk, dl, hb); operands requested two MFMAs ahead (one V fragment, two P fragments in flight): the
        // register budget has no room for a deeper ring (128 accumulators + 72 of Q^T out of 256)
        s16x8 af[2], pfr[3];
        __builtin_amdgcn_sched_barrier(0);
        af[0] = lda(0);
        pfr[0] = ldp(0, 0);
        pfr[1] = ldp(1, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int step = i >> 2, hb = i & 3;               // step = kk * 2 + dl
            __builtin_amdgcn_sched_barrier(0);
            if (hb == 0 && step + 1 < 4) af[(step + 1) & 1] = lda(step + 1);
            if (i + 2 < 16) pfr[(i + 2) % 3] = ldp((i + 2) & 3, (i + 2) >> 3);
            __builtin_amdgcn_sched_barrier(0);
            const int a = (step & 1) * 4 + hb;
            acc[a] = mfma32<BF16>(af[step & 1], pfr[i % 3], acc[a]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto fill_only = [&](int t) {
        const Pieces8 rows3 = tile_top(t);
        const uint32_t nslot = cx.lds_base + (uint32_t)(((t + k8Lead) & (k8Slots - 1)) * k8SlotBytes);
#pragma unroll
        for (int i = 0; i < 5; ++i) issue_piece8(cx, rows3, nslot, i);
    };
    const int hbw = wave >> 1;                                  // exchange-buffer coordinates of this wave's P^T pieces: consumer lane
    const int lc = (g & 1) * 32 + (wave & 1) * 16 + h16;        // (kg = g & 1, c32 = 16 (w & 1) + h16), half g >> 1 of its 16 bytes
    uint8_t *const pdst = pbuf + (hbw * 2 * 64 + lc) * 16 + (g >> 1) * 8;
    if (!wave_active) {                                        // no heads of its own: P = 0 for its block, DMA share and P.V slice as usual
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) *(uint2 *)(pdst + kb * 1024) = uint2{0u, 0u};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int t = t_begin; t < t_end; ++t) {
            fill_only(t);
            asm volatile("s_barrier" ::: "memory");              // barrier B
            pv(t);
        }
    }

    // ---- S^T[key, head] = K . Q^T: 18 k-steps of 32 dims x 2 key blocks of 16, operand ring kAhead deep, a DMA piece every 7 MFMAs
    auto qk = [&](int t, const Pieces8 &rows3, f32x4 &s0, f32x4 &s1) {
        const uint8_t *buf = lds + (t & (k8Slots - 1)) * k8SlotBytes;
        const uint32_t nslot = cx.lds_base + (uint32_t)(((t + k8Lead) & (k8Slots - 1)) * k8SlotBytes);
        const uint8_t *abase = buf + h16 * kNopeStride + g * 16;
        const uint8_t *rbase = buf + k8T * kNopeStride + h16 * kRopeStride;
        auto lda = [&](int step) -> s16x8 {                    // step = ks * 2 + kb; key 16 kb + h16, dims 32 ks + 8 g .. +8
            const int ks = step >> 1, kb = step & 1;
            if (ks < 16) return *(const s16x8 *)(abase + kb * 16 * kNopeStride + ks * 64);
            return *(const s16x8 *)(rbase + kb * 16 * kRopeStride + ((((ks - 16) * 4 + g) ^ (h16 & 7)) << 4));
        };
        constexpr int kAhead = 2, kRing = kAhead + 1;
        s16x8 af[kRing];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pre = 0; pre < kAhead; ++pre) af[pre] = lda(pre);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            __builtin_amdgcn_sched_barrier(0);
            if (step + kAhead < 36) af[(step + kAhead) % kRing] = lda(step + kAhead);
            __builtin_amdgcn_sched_barrier(0);
            if (step == 0) mfma16_first<BF16>(s0, af[0], qf[0]);
            else if (step == 1) mfma16_first<BF16>(s1, af[1], qf[0]);
            else if (step & 1) mfma16_acc<BF16>(s1, af[step % kRing], qf[step >> 1]);
            else mfma16_acc<BF16>(s0, af[step % kRing], qf[step >> 1]);
            if (step % 7 == 3) issue_piece8(cx, rows3, nslot, step / 7);       // steps 3, 10, 17, 24, 31 -> pieces 0..4
        }
        mfma16_settle(s0, s1);
        __builtin_amdgcn_sched_barrier(0);
    };

    constexpr float kGuard = BF16 ? 64.0f : 11.0f;
#ifdef MLA8_TIMING
    uint64_t tm[6] = {0, 0, 0, 0, 0, 0}, c0 = 0, c1;
    const uint64_t t_entry = __builtin_amdgcn_s_memtime(), r_entry = __builtin_amdgcn_s_memrealtime();
#define MLA8_TICK(i) c1 = __builtin_amdgcn_s_memtime(); tm[i] += c1 - c0; c0 = c1;
#else
#define MLA8_TICK(i)
#endif
    if (wave_active) {
        for (int t = t_begin; t < t_end; ++t) {
#ifdef MLA8_TIMING
            c0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            MLA8_TICK(0)
#endif
            const Pieces8 rows = tile_top(t);
            MLA8_TICK(1)
            f32x4 s0, s1;
            qk(t, rows, s0, s1);
            MLA8_TICK(2)
            // lane (h16, g) holds head h16, keys 16 kb + 4 g + i.  Only the tile that crosses seq_len needs the mask.
            if ((t + 1) * k8T > seq_len) {
                asm volatile("" ::: "memory");
                const int kbase = t * k8T + 4 * g;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (kbase + i >= seq_len) s0[i] = -INFINITY;
                    if (kbase + 16 + i >= seq_len) s1[i] = -INFINITY;
                }
            }
            float tmax = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
            if (t == t_begin) {                           // the softmax reference of this head: first tile's maximum over all 32 keys
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                m_run = tmax * cs;                             // sm_scale > 0: max commutes with the scaling
            } else if (__any(tmax * cs > m_run + kGuard)) {
                *(uint32_t *)(lds + k8FlagOff) = 1;
            }
            const float nm = (m_run == -INFINITY) ? 0.f : -m_run;
            float e[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[i], cs, nm));
                e[4 + i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], cs, nm));
            }
            l_run += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
            *(uint2 *)(pdst) = uint2{pack2<BF16>(e[0], e[1]), pack2<BF16>(e[2], e[3])};
            *(uint2 *)(pdst + 1024) = uint2{pack2<BF16>(e[4], e[5]), pack2<BF16>(e[6], e[7])};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MLA8_TICK(3)
            asm volatile("s_barrier" ::: "memory");               // barrier B: P^T(t) complete
            MLA8_TICK(4)
            pv(t);
            MLA8_TICK(5)
        }
        l_run += __shfl_xor(l_run, 16, 64);                    // the four key groups of a head
        l_run += __shfl_xor(l_run, 32, 64);
    }
#ifdef MLA8_TIMING
    if (lane == 0 && blockIdx.x < 64) {          // [own vmcnt wait, barrier A + addresses, QK^T, softmax + publish, barrier B, P.V] per tile, then loop totals
        float *dbg = (float *)p.fix_flags + 2048 + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) dbg[i] = (float)tm[i] / (float)max(1, t_end - t_begin);
        dbg[6] = (float)(__builtin_amdgcn_s_memtime() - t_entry);
        dbg[7] = (float)(__builtin_amdgcn_s_memrealtime() - r_entry);
    }
    const uint64_t r_loop_end = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // fills issued past the last tile
    __syncthreads();
    const bool flagged_local = *(volatile uint32_t *)(lds + k8FlagOff) != 0;
    if (threadIdx.x == 0 && flagged_local) p.fix_flags[b * p.kv_heads + kvh] = p.fix_epoch;

    // ---- epilogue: acc[dl * 4 + hb][4 rg + i] = O^T[d][head 32 hb + c32], d = 128 (w >> 1) + 64 (rg >> 1) + 16 (2 (w & 1) + dl) + 8 (rg & 1) + 4 kg + i;
    // the softmax statistics of a head live in the wave that owns it and reach the others through LDS
    float *lmb = (float *)pbuf;                                // [0..127] l, [128..255] m (all P.V reads are behind the barrier above)
    if (g == 0) {
        lmb[wave * 16 + h16] = wave_active ? l_run : 0.f;
        lmb[128 + wave * 16 + h16] = wave_active ? m_run : -INFINITY;
    }
    __syncthreads();
    // Which piece of its sequence this workgroup is, and where its partial goes, is formed HERE and not at the top: nothing of it stays
    // live across the tile loop.  Partial slot of head `headx`: one row per (item, head of the group) in the planned form, per (sequence,
    // head, split) otherwise.
    int nsplits, pmul;
    int64_t pbase;
    if constexpr (PLAN) {
        nsplits = plan_item(p.plan, (long long)p.batch * p.kv_heads, (int)blockIdx.x).n;
        pmul = 1, pbase = ((int64_t)blockIdx.x - kvh) * p.group;
    } else {
        nsplits = p.num_splits;
        pmul = p.num_splits, pbase = (int64_t)b * p.q_heads * p.num_splits + ((blockIdx.x >> 3) / head_blocks) % p.num_splits;
    }
    auto pslot = [&](int headx) -> int64_t { return pbase + (int64_t)headx * pmul; };
    const bool finals = nsplits == 1;                          // this workgroup writes output rows itself (no merge launch)
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
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) {                           // static accumulator indices: keep this loop unrolled
        if (hblk * 128 + hb * 32 >= p.group) continue;
#pragma unroll
        for (int dl = 0; dl < 2; ++dl)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int col = (rg >> 1) * 32 + dl * 16 + (rg & 1) * 8 + 4 * kg;
                const f32x16 &a = acc[dl * 4 + hb];
                *(f32x4 *)(tile + c32 * kEpiRow + col * 4) = f32x4{a[4 * rg + 0], a[4 * rg + 1], a[4 * rg + 2], a[4 * rg + 3]};
            }
        // wave-private tile: LDS operations of one wave complete in order, no barrier
        f32x4 o8[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) o8[it] = *(const f32x4 *)(tile + (it * 4 + (lane >> 4)) * kEpiRow + (lane & 15) * 16);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int hl = it * 4 + (lane >> 4), ch = lane & 15;
            const int hgx = hblk * 128 + hb * 32 + hl;
            const int d = dcol0 + (ch >> 3) * 64 + (ch & 7) * 4;
            const int headx = kvh * p.group + min(hgx, p.group - 1);
            if (hgx >= p.group) continue;
            if (finals) {
                const float l_h = lmb[hb * 32 + hl];
                const float inv = l_h > 0.f ? 1.f / l_h : 0.f;
                const uint32_t w0 = (uint32_t)cvt_out<BF16>(o8[it][0] * inv) | ((uint32_t)cvt_out<BF16>(o8[it][1] * inv) << 16);
                const uint32_t w1 = (uint32_t)cvt_out<BF16>(o8[it][2] * inv) | ((uint32_t)cvt_out<BF16>(o8[it][3] * inv) << 16);
                *(uint2 *)(p.out + (int64_t)b * p.o_sb + (int64_t)headx * p.o_sh + d) = uint2{w0, w1};
            } else {
                *(f32x4 *)(p.ws_o + pslot(headx) * kDN + d) = o8[it];
            }
        }
        if (!finals && wave == 0 && lane < 32) {               // softmax statistics of the block's 32 heads: lane = head
            const int hgx = hblk * 128 + hb * 32 + lane;
            if (hgx < p.group) {
                const int64_t idx = pslot(kvh * p.group + hgx);
                p.ws_ml[idx * 2 + 0] = lmb[128 + hb * 32 + lane];
                p.ws_ml[idx * 2 + 1] = lmb[hb * 32 + lane];
            }
        }
    }
#ifdef MLA8_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {      // absolute 100 MHz stamps per workgroup: [start, Q^T resident (loop entry), loop end, exit]
        uint64_t *w = (uint64_t *)((float *)p.fix_flags + 2048 + 64 * 8 * 8) + (size_t)blockIdx.x * 4;
        w[0] = r_start, w[1] = r_entry, w[2] = r_loop_end, w[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace

void launch_mla_wide8(const MlaParams &p, int dtype, long long units, hipStream_t st)
{
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, k8Lds);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, k8Lds);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, k8Lds);
        (void)hipFuncSetAttribute((const void *)mla_decode_wide8_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, k8Lds);
    }
    const int head_blocks = (p.group + 127) / 128;
    const long long seqs = p.plan ? 0 : units / p.num_splits;  // (sequence, kv head) pairs, 8 per grid row of XCDs
    // planned form: `units` = the upper bound of the work list (plan_items_max); workgroups behind the list's end leave at once
    dim3 grid(p.plan ? (unsigned)units : (unsigned)(((seqs + 7) / 8) * 8 * p.num_splits * head_blocks));
    if (p.plan) {
        if (dtype == MI_DTYPE_BF16) mla_decode_wide8_kernel<true, true><<<grid, 512, k8Lds, st>>>(p);
        else mla_decode_wide8_kernel<false, true><<<grid, 512, k8Lds, st>>>(p);
    } else {
        if (dtype == MI_DTYPE_BF16) mla_decode_wide8_kernel<true, false><<<grid, 512, k8Lds, st>>>(p);
        else mla_decode_wide8_kernel<false, false><<<grid, 512, k8Lds, st>>>(p);
    }
}

}  // namespace mi_sgl
