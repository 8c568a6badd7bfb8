// Grouped INT8 expert GEMMs of fused_deep_moe for gfx950 (v_mfma_i32_16x16x64_i8).
// Replaces the "catlass/act" AscendC GEMM templates the reference fuses into one MIX kernel
// (csrc/deepep/ops/op_kernel/fused_deep_moe.h:336-427): GEMM1 int8[R,H] x int8[H,2I] -> i32 with the per-token dequant +
// SwiGLU epilogue (ops/utils/op_kernel/operator/epilogue/block/block_epilogue_per_token_dequant_swiglu.h:250-269),
// the per-row requantisation (.../gemm/kernel/grouped_matmul_slice_m_per_token_dequant_swiglu_quant_multistage_workspace.h:199-265)
// and GEMM2 int8[R,I] x int8[I,H] with per-token x per-channel dequant to bf16.
//
// MI355X design: weights are consumed as [expert][N][K] (K contiguous), activations as [row][K]: both MFMA operands are
// 16-byte K-slices.  What limits a grouped GEMM of this shape on this part is the operand stream into the CUs (L2 -> LDS
// saturated near 47 GB/s per CU / 12 TB/s per chip in every variant tried), so the workgroup tile is as large as the LDS
// allows: 256(M) x 256(N) x 64(K bytes) = 3.9 KB of operands per int8 MOP, 16 waves in a 4 x 4 grid of 64 x 64 wave tiles
// (4 x 4 MFMA tiles = 64 accumulator registers, ~110 VGPRs -> 4 waves per SIMD, which is what hides the LDS latency).
// Tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 rows x 64 B per wave-instruction, no staging registers),
// XOR swizzle chunk ^ ((row >> 2) & 3) applied on the source side so the 16-row ds_read_b128 footprints are
// conflict-free, into a ring of 4 stages (4 x 32 KB), three k-tiles in flight, one barrier per k-tile.
// History: 128x128x128 tiles staged through VGPRs with one tile of look-ahead ran at 1.22 POPS (every k-tile waited out a
// global-load latency); 128x256 with a 3-stage DMA ring 1.44; + dedicated loader waves 1.73 (MFMA waves then waited on the
// loaders: bandwidth, not issue); this version trades the loaders for a third less traffic.
// A SwiGLU fusion tile is 128 columns = 64 gate | 64 up (weights pre-permuted, reference test_fused_deep_moe.py:75-86); a wave
// takes 32 gate columns and the matching 32 up columns, so the dequant + SwiGLU epilogue needs no exchange.
// Expert row ranges come from the device-side cumulative counts, so the same launch serves low-latency (no host sync) and
// normal mode; idle tile slots exit at once.
// Bound: operand stream / MFMA int8 for prefill-size groups (2*M*N*K ops), HBM (weights once: L*N*K bytes) for decode.
#include "device_once.h"
#include "ep_common.h"

namespace mi_ep {

constexpr int BN = 256, BK = 64;
// ring depth of the operand stages: 4 x 64-byte k-tiles (4 x 32 KB for the 256-row tile, 4 x 20 KB for the 64-row tile), 3 x 128-byte
// k-tiles (3 x 40 KB) for the decode tile when K allows it.  (A 5-deep ring for the 256-row tile -- all 160 KB of LDS, four stages in
// flight -- measured the same as 4: GEMM1 1.08 ms, GEMM2 0.74 ms at C5.  The wait at the k-tile barrier is not a ring-depth effect.
// Two k-tiles per barrier (the pair being multiplied + the pair in flight): GEMM1 1.085, GEMM2 0.705 -- within the box-to-box
// noise; the same with a 5-deep ring and three stages in flight: 1.14 / 0.74, i.e. MORE requests in flight make it slower.  What
// the barrier waits for is the operand stream itself (L2 misses on activation tiles that all 8 XCDs fetch), not latency.
// Starting every tile's walk over K at a different k-tile (exact: integer accumulation) -- which spreads the memory channels a
// lockstep launch hits and helped the per-head mla_preprocess kernel -- is WORSE here: GEMM1 1.07 -> 1.14 ms, GEMM2 0.73 -> 0.80 ms
// (decode tile: 170 -> 175 us, 115 -> 110 us): the tiles that share an operand tile want to read the same k-tile at the same time.)
// 256-row tile with 128-byte k-tiles (round 2c): a stage is 64 KB, the ring two stages deep -- one k-tile in flight.  With 64-byte k-tiles
// every LDS-DMA request used half of a 128-byte line and the other half was fetched again a stage later (the 32 KB L1 does not
// hold a stage): GEMM1 1.14 -> 1.02 ms, GEMM2 0.786 -> 0.726 ms at C5, same box.  A ring of five HALF stages (A or B of a k-tile,
// 32 KB each: A(kt+1), B(kt+1), A(kt+2) in flight) was built on top of it, bit-exact, and is slower again: 1.08 / 0.77 ms.
// Ablations of that kernel at C5 (timing only, wrong products): without the refills inside the k-loop GEMM1 takes 0.79 ms and GEMM2
// 0.56 ms (the operand stream costs 23 % although it is asynchronous); with 5 instead of 8 operand fragments read from LDS per
// k-step 0.97 / 0.74 ms (the LDS read volume is NOT the limiter: larger wave tiles would buy <= 5 %).  Per k-tile a wave spends
// ~40 % of its time at the barrier even without refills: four waves share a SIMD's MFMA pipe, and after the barrier every wave first
// waits out its operand reads.  Carrying the next k-step's fragments across the barrier needs a second fragment set (32 VGPRs) next
// to 64 accumulators inside the 128-register budget of 16 waves per CU.
template <int BKT, int MT> struct RingDepth { static constexpr int value = BKT == 128 ? (MT == 4 ? 2 : 3) : 4; };
constexpr int ring_bytes(int bkt, int mt) { return (bkt == 128 ? (mt == 4 ? 2 : 3) : 4) * (64 * mt + 256) * bkt; }
constexpr int kGemmThreads = 1024;
constexpr int kEpiRowBytes = 144;      // epilogue transpose tile: 128-byte rows + 16 B (see the epilogue)
constexpr int kRqCols = 16;            // requantising GEMM1: words per row in the exchange line (one per 256-column tile: two_i <= 4096)
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const int8_t *a;          // [M_cap, K]; with a_rows: the base the row offsets count from
    const uint32_t *a_rows;   // null, or [M_cap] byte offsets of the activation rows from `a` (16-byte aligned): the rows are read where a
                              // dispatch staged them (one row per TOKEN, shared by its K selections) instead of from a gathered copy
    const float *a_scale;     // [M_cap]
    const int8_t *w;          // [L, N, K]
    const float *w_scale;     // [L, N]
    const int32_t *cum;       // inclusive cumulative row counts; expert e ends at cum[(e + 1) * cum_stride - 1]
    int cum_stride, L, M_cap, K, N;
    void *out;                // mode 0: float [M_cap, N/2]; mode 1: bf16 [M_cap, N]; mode 2: unused
    // mode 2 (GEMM2 fused with the combine push): row r goes to slot t*K+k of rank src, (src, t, k) = src_idx[3 r ..]
    const int32_t *src_idx;
    PeerPtrs dsts;
    size_t slot_stride;
    int topk, W;
    Parity par;               // ping-pong half of the combine window (device-resident epoch, see ep_common.h)
    int slot_rows;            // rows one combine region holds: (t, k) outside it are dropped
    int cols_padded;          // 0, or the column-tile count rounded up to a multiple of 8 (XCD-consistent column tiles, see the kernel)
    int small_last;           // an expert's last row block of <= 64 / <= 128 rows runs as a 64- / 128-row tile (MI_GEMM_SMALL_LAST=0: as a 256-row tile)
    // mode 3 (GEMM1 with the per-row requantisation in its epilogue, see gemm_tile): int8 [M_cap, N/2] + float [M_cap] out, and the words the
    // column tiles of a row block meet at -- all zero when the launch starts (one memset per call: mi_ep_moe_requant_words)
    int8_t *q_out;
    float *q_scale;
    uint32_t *rq_rowmax;      // [M_cap][kRqCols] exchange lines: word c of row r = bits of column tile c's max |v| of that row | 0x80000000
    uint32_t *rq_tickets;     // [8] workgroups started per XCD (see grouped_gemm_i8_kernel)
    int rq_xcds;              // XCDs the workgroups are dealt to round-robin (verified at start-up: mi_ep_moe_probe_xcds); 1 = one ticket for all
    int32_t *status;
    uint64_t timeout_ticks;
};

// position (16-B units) of k-chunk `chunk` inside row `row` of a [rows][BKT B] tile.  A ds_read_b128 of 16 consecutive rows at one chunk
// then covers all 64 banks: 64-B rows -- 4 rows span the banks, the position rotates every 4 rows; 128-B rows -- 2 rows span the banks, the
// position rotates every 2 rows.
template <int BKT> __device__ __forceinline__ int swz_pos(int row, int chunk)
{
    return BKT == 64 ? (chunk ^ ((row >> 2) & 3)) : (chunk ^ ((row >> 1) & 7));
}
template <int BKT> __device__ __forceinline__ int swz(int row, int chunk) { return row * BKT + (swz_pos<BKT>(row, chunk) << 4); }

// LDS-DMA through inline asm (the compiler then places no vmcnt wait of its own; ordering is the explicit vmcnt below).
// Lane l moves 16 B from its own address to dst + 16 l; M0 carries the wave-uniform LDS destination.
// Address = wave-uniform base (SGPR pair) + 32-bit lane offset: one VGPR per operand stream instead of a 64-bit pointer per lane.
__device__ __forceinline__ void dma16(uint32_t dst, const void *sbase, uint32_t voff)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(voff), "s"(sbase) : "memory");
}

#ifdef GEMM_TIMING
__device__ float g_gemm_dbg[256];
#endif

// MT = MFMA row tiles per wave: 4 -> 256-row workgroup tile (prefill-size groups), 1 -> 64-row tile (decode-size groups:
// the weights stream once either way, the small tile just stops multiplying padding).
// BKT = bytes of K per stage: 64, or 128 for the decode tile -- a decode-size group is a pure weight stream, and with 64-byte k-tiles
// every request touches half a 128-byte line of a weight row (the other half comes a stage later): 3.8-5.1 TB/s.  A deeper ring did not
// move that (7 x 20 KB measured the same as 4 x 20 KB), whole lines per request did (GEMM1 184 -> 169 us, GEMM2 124 -> 114 us at 128
// tokens x 32 experts; a 2-deep ring of 128-byte k-tiles, two workgroups per CU, gave 185 / 121 us).
// one (expert, m-tile) x 256-column tile; `slot` = index of the tile in (expert, row block) order.  Returns false when the slot lies
// past the last tile (workgroup-uniform).
// Which (expert, row block) is tile slot `tile_slot`?  Row blocks are 256 rows (the prefill tile) or 64 (the decode tile); 64 experts per
// step: lane i reads the end of expert i, a wave scan of the per-expert tile counts locates the slot.  (A serial walk over the experts --
// one dependent scalar load each -- cost up to ~15 us per workgroup for the last experts, a third of a GEMM2 tile.)  Returns false when
// the slot lies past the last tile (workgroup-uniform).
template <int BM>
__device__ __forceinline__ bool find_tile(const GemmArgs &p, int tile_slot, int end_first, int &e_out, int &row0_out, int &rows_out)
{
    const int lane = threadIdx.x & 63;
    int e = -1, row0 = 0, rows = 0;
    {
        int slot = tile_slot, start = 0;                    // wave-uniform
        for (int base = 0; base < p.L && e < 0; base += 64) {
            const int i = base + lane;
            // the ends of the first 64 experts were loaded once per workgroup (grouped_gemm_i8_kernel): one dependent global load
            // less in front of every tile
            const int end = base == 0 ? end_first : (i < p.L ? p.cum[(i + 1) * p.cum_stride - 1] : 0);
            int prev = __shfl_up(end, 1, 64);
            if (lane == 0) prev = start;
            const int cnt = i < p.L ? end - prev : 0;
            const int tiles = (cnt + BM - 1) / BM;
            int incl = tiles;                               // inclusive scan of the tile counts
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int n = __shfl_up(incl, off, 64);
                if (lane >= off) incl += n;
            }
            const unsigned long long hit = __ballot(i < p.L && incl > slot);
            if (hit) {
                const int l = __builtin_ctzll(hit);
                const int before = __shfl(incl - tiles, l, 64);
                const int s0 = __shfl(prev, l, 64), c = __shfl(cnt, l, 64);
                e = base + l;
                row0 = s0 + (slot - before) * BM;
                rows = min(BM, c - (slot - before) * BM);
            } else {
                const int last = min(63, p.L - 1 - base);
                slot -= __shfl(incl, last, 64);
                start = __shfl(end, last, 64);
            }
        }
    }
    e_out = __builtin_amdgcn_readfirstlane(e);
    row0_out = __builtin_amdgcn_readfirstlane(row0);
    rows_out = __builtin_amdgcn_readfirstlane(rows);
    return e_out >= 0;
}

// one (expert, row block of `rows` <= 64 MT rows starting at row0) x 256-column tile
template <int MODE, int MT, int BKT>
__device__ __forceinline__ void gemm_tile(const GemmArgs &p, int e, int row0, int rows, int col_tile, uint8_t *lds)
{
    constexpr bool SWIGLU = MODE == 0 || MODE == 3;   // GEMM1: fusion tiles of 64 gate | 64 up columns; MODE 3 also requantises the rows
    constexpr int BM = 64 * MT;
    constexpr int kStages = RingDepth<BKT, MT>::value;
    constexpr int kStageBytes = (BM + BN) * BKT;
    constexpr int kPieceRows = 1024 / BKT;            // rows one DMA instruction (64 lanes x 16 B) covers
    constexpr int kChunks = BKT / 16;                 // 16-B chunks per row
    constexpr int kAPieces = BM / kPieceRows;         // DMA instructions for the A tile of a stage
    constexpr int kAPerWave = (kAPieces + 15) / 16;   // A pieces per wave: wave w issues pieces w, w + 16, ... below kAPieces
    constexpr int kBPerWave = BN / kPieceRows / 16;   // B pieces every wave issues per stage
    static_assert((kAPieces <= 16 || kAPieces % 16 == 0) && kBPerWave >= 1, "DMA plan");
#ifdef GEMM_TIMING
    const uint64_t t_entry = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c16 = lane & 15;
    // (Round 4, measured and dropped: wm = wave >> 2 -- the four row quarters of a column strip on one SIMD -- plus skipping the MFMAs of row
    //  tiles without rows, so that an expert's last, partly filled tile loads all four SIMDs with what rows it has.  The mapping alone is
    //  neutral; a wave-uniform test in front of the MFMA groups cost full tiles 7.5 %; with the k-loop duplicated -- plain for full tiles,
    //  predicated for remainder tiles -- multinomial row counts gained 2 % in GEMM1 and lost 1-3 % elsewhere (tools/time_gemm.py, "ragged").
    //  A remainder tile costs what a full one costs because its weight tile streams all the same: the k-tile time is the operand stream's.
    //  Round 5: an expert's last row block of <= 128 rows runs as a 64- or 128-row tile instead -- grouped_gemm_i8_kernel.)
    const int wm = wave & 3, wn = wave >> 2;
    const int n0 = col_tile * BN;
    const int8_t *wbase = p.w + (size_t)e * p.N * p.K;
    const int8_t *abase = p.a_rows ? p.a : p.a + (size_t)row0 * p.K;

    // ---- DMA plan: one instruction moves kPieceRows rows x BKT bytes (1 KB); wave w issues A piece w (if there is one) and B pieces
    // w, w + 16, ...
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);
    uint32_t offA[kAPerWave], offB[kBPerWave];    // lane offsets from the wave-uniform bases abase / wbase
    {
        const int row = kPieceRows * wave + lane / kChunks;
        const int chunk = swz_pos<BKT>(row, lane % kChunks);                      // swizzle on the source side (an involution)
#pragma unroll
        for (int j = 0; j < kAPerWave; ++j) {                                     // rows past the group: any valid row
            const int r = min(row + j * 16 * kPieceRows, rows - 1);
            offA[j] = (p.a_rows ? p.a_rows[row0 + r] : (uint32_t)r * (uint32_t)p.K) + chunk * 16;
        }
#pragma unroll
        for (int j = 0; j < kBPerWave; ++j) {
            const int brow = row + j * 16 * kPieceRows;                           // 128 rows further: the same swizzle term
            offB[j] = (uint32_t)min(n0 + brow, p.N - 1) * (uint32_t)p.K + chunk * 16;
        }
    }
    const int nk = p.K / BKT;
    auto issue_stage = [&](int kt) {
        const int kc = min(kt, nk - 1) * BKT;             // past the end: a harmless refill keeps the vmcnt arithmetic uniform
        const uint32_t sbase = lds_base + (uint32_t)((kt % kStages) * kStageBytes + wave * 1024);
#pragma unroll
        for (int j = 0; j < kAPerWave; ++j)
            if (wave + 16 * j < kAPieces) dma16(sbase + (uint32_t)(j * 16 * 1024), abase + kc, offA[j]);     // wave-uniform
#pragma unroll
        for (int j = 0; j < kBPerWave; ++j) dma16(sbase + (uint32_t)(BM * BKT + j * 16 * 1024), wbase + kc, offB[j]);
    };

    i32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};

    // LDS byte offsets of this lane's operand fragments inside a stage (k-step ks = 64 bytes of the row: chunk 4 ks + g)
    constexpr int kSteps = BKT / 64;
    int aoff[kSteps][MT], boff[kSteps][4];
#pragma unroll
    for (int ks = 0; ks < kSteps; ++ks) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) aoff[ks][mt] = swz<BKT>(wm * 16 * MT + mt * 16 + c16, 4 * ks + g);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int brow;
            if (SWIGLU) brow = (wn >> 1) * 128 + (nt >> 1) * 64 + (wn & 1) * 32 + (nt & 1) * 16 + c16;   // nt 0,1 gate; 2,3 up
            else brow = wn * 64 + nt * 16 + c16;
            boff[ks][nt] = BM * BKT + swz<BKT>(brow, 4 * ks + g);
        }
    }

#ifdef GEMM_TIMING
    uint64_t tw = 0, tc = 0, c0 = __builtin_amdgcn_s_memtime(), c1;
    const uint64_t t_loop = c0;
#endif
#pragma unroll
    for (int st = 0; st < kStages - 1; ++st) issue_stage(st);
    for (int kt = 0; kt < nk; ++kt) {
#ifdef GEMM_TIMING
        c0 = __builtin_amdgcn_s_memtime();
#endif
        // own pieces of stage kt landed; those of the kStages - 2 younger stages may still fly (kBPerWave per stage, one more for
        // the waves with an A piece)
        if (kAPieces >= 16 || wave < kAPieces) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kBPerWave + kAPerWave) * (kStages - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kBPerWave * (kStages - 2)) : "memory");
        __syncthreads();                                    // stage kt complete; the slot of stage kt-1 is free
#ifdef GEMM_TIMING
        c1 = __builtin_amdgcn_s_memtime(); tw += c1 - c0; c0 = c1;
#endif
        issue_stage(kt + kStages - 1);
        const uint8_t *buf = lds + (kt % kStages) * kStageBytes;
        if constexpr (MT == 4) {
            // Operand fragments two MFMA groups ahead of their use.  Left to itself the compiler emitted `ds_read -> s_waitcnt lgkmcnt(0) ->
            // 4 MFMAs` eight times per k-tile: every group of four MFMAs (64 cycles) waited out a full LDS read, and only the other three
            // waves of the SIMD covered for it.  Order per k-tile: the four column fragments of k-step 0 and the first two row
            // fragments; then per row fragment i: request fragment i + 2, multiply fragment i.  sched_barrier pins the order, the waits
            // are the compiler's (LDS returns in order).
            constexpr int NA = kSteps * MT;
            i32x4 bfr[4], ar[3];          // ONE set of column fragments: the set of k-step 1 replaces that of k-step 0 register by register,
                                          // each right behind the last MFMA that read it (two sets spilled 22 registers in the epilogue)
            auto load_a = [&](int idx) { return *(const i32x4 *)(buf + aoff[idx / MT][idx % MT]); };
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bfr[nt] = *(const i32x4 *)(buf + boff[0][nt]);
            ar[0] = load_a(0);
            ar[1] = load_a(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int idx = 0; idx < NA; ++idx) {
                const int mt = idx % MT;
                const bool last_of_step = kSteps == 2 && idx == MT - 1;
                if (idx + 2 < NA) ar[(idx + 2) % 3] = load_a(idx + 2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bfr[nt], ar[idx % 3], acc[mt][nt], 0, 0, 0);      // operands swapped, see below
                    if (last_of_step) {
                        __builtin_amdgcn_sched_barrier(0);
                        bfr[nt] = *(const i32x4 *)(buf + boff[kSteps - 1][nt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int ks = 0; ks < kSteps; ++ks) {
            i32x4 af[MT], bf[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt] = *(const i32x4 *)(buf + aoff[ks][mt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nt] = *(const i32x4 *)(buf + boff[ks][nt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    // operands swapped: the MFMA produces the TRANSPOSED block, i.e. lane (g, c16) holds C[row mt*16 + c16][columns
                    // nt*16 + 4g .. +3] -- four consecutive columns of one row, which the epilogue packs and writes as one piece
                    acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
#ifdef GEMM_TIMING
        c1 = __builtin_amdgcn_s_memtime(); tc += c1 - c0;
#endif
    }
#ifdef GEMM_TIMING
    if (lane == 0 && blockIdx.x == 3 && blockIdx.y < 4) {
        g_gemm_dbg[(blockIdx.y * 16 + wave) * 4 + 0] = (float)tw / nk;
        g_gemm_dbg[(blockIdx.y * 16 + wave) * 4 + 1] = (float)tc / nk;
        g_gemm_dbg[(blockIdx.y * 16 + wave) * 4 + 2] = (float)(t_loop - t_entry);
    }
    const uint64_t t_epi = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the refills issued past the last k-tile

    // ---- epilogue: lane holds C[row = wm*16*MT + mt*16 + c16][n-tile nt, columns 4g .. 4g+3] (transposed MFMA blocks, see the k-loop).
    // Written straight from the accumulators a wave-store would cover 16 rows x 32 B; instead every wave passes its tile through LDS (the
    // operand ring is free now) and stores whole 128-byte rows, 16 B per lane.  Row stride 144 B.  A lane dequantises four consecutive
    // columns of a row: one 16-byte load of their weight scales per n-tile, one activation scale per row tile, hardware bf16 packing
    // (v_cvt_pk_bf16_f32), and ONE LDS write per (row tile, n-tile).  (Round 2b held the untransposed block -- 4 rows x 1 column per
    // lane: 2-byte LDS writes, a software bf16 rounding of ~7 VALU operations per element and 16 activation scales per lane; on the
    // 16-lane SIMDs that VALU work, 4 cycles per wave instruction and four waves per SIMD, was most of the epilogue's 17.5k cycles.)
    const float *ws = p.w_scale + (size_t)e * p.N + n0;
    constexpr int kRowBytes = kEpiRowBytes, kWaveRows = 16 * MT;
    __syncthreads();                                        // every wave is done reading the ring (and nothing is in flight)
    uint8_t *tile = lds + wave * (kWaveRows * kRowBytes);
    const int f = wn >> 1, h = wn & 1;                      // MODE 0: fusion tile f (128 columns: 64 gate | 64 up), half h
    const bool wave_cols_ok = SWIGLU ? (n0 + f * 128 < p.N) : (n0 + wn * 64 < p.N);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 wsv[4];                                           // weight scales of this lane's columns: n-tile nt, columns 4g .. 4g+3
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int col;                                            // first of the four columns, relative to n0
        if (SWIGLU) col = f * 128 + (nt >> 1) * 64 + h * 32 + (nt & 1) * 16 + 4 * g;      // nt 0,1 gate; 2,3 up
        else col = wn * 64 + nt * 16 + 4 * g;
        wsv[nt] = wave_cols_ok ? *(const f32x4 *)(ws + col) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (MODE == 3) {
        // ---- GEMM1 with the per-row requantisation in the epilogue (the reference requantises in GEMM1's epilogue too:
        // block_epilogue_per_token_dequant_swiglu.h:250-269, ...swiglu_quant_multistage_workspace.h:199-265).  A row's maximum spans all N / 256
        // column tiles of its row block, i.e. N / 256 workgroups: each keeps its SwiGLU values in registers (32 per lane), posts its rows'
        // maxima into the rows' exchange lines (one word per column tile), waits (bounded) until the other column tiles' words are there,
        // then quantises from the registers with the final maxima -- the same values, the same formula as rowquant_kernel, so the
        // same bits -- and writes int8 rows instead of fp32 ones.  No fp32 intermediate crosses HBM and there is no rowquant launch.
        // Forward progress: the workgroups that share a row block were formed at their start (grouped_gemm_i8_kernel) and walk the same tile
        // slots in the same order, so a waiter only ever waits for workgroups that are already running.
        float *part = (float *)lds;                          // [4 column quarters][BM] row maxima of this workgroup's waves
        float *invs = part + 4 * BM;                         // [BM] 1 / row maximum (0 for an all-zero row)
        uint8_t *qt = lds + 8192;                            // [BM] rows x 128 int8 (+ 16 B: kEpiRowBytes), the workgroup's output tile
        f32x4 v[MT][2];
        float rmax[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int rl = mt * 16 + c16;
            const float as = p.a_scale[(size_t)row0 + min(wm * kWaveRows + rl, rows - 1)];
            float m = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gate = ((float)acc[mt][nt][r] * wsv[nt][r]) * as;
                    const float up = ((float)acc[mt][nt + 2][r] * wsv[nt + 2][r]) * as;
                    v[mt][nt][r] = up * (gate * __builtin_amdgcn_rcpf(1.0f + __expf(-gate)));      // (as MODE 0 below)
                    m = fmaxf(m, fabsf(v[mt][nt][r]));
                }
            }
            // over the four 16-lane rows of the wave (lanes c16, c16 + 16, c16 + 32, c16 + 48 hold the same matrix row)
            typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
            u32x2v sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            rmax[mt] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        if (g == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) part[wn * BM + wm * kWaveRows + mt * 16 + c16] = rmax[mt];
        }
        __syncthreads();
        if (tid < BM) {
            // Thread r owns row r of the block.  Its word of the row's exchange line -- 16 words, one per column tile -- is the maximum of this
            // workgroup's 128 columns with the SIGN bit set: |v| >= 0, so the bit is free, and a word that is still zero has not been posted
            // (the line is its own flag: one device-scope store, no counter, no second round trip).  Then it polls the line until every
            // column tile's word is there.
            float amax = 0.f;
            if (tid < rows) {
                const float m = fmaxf(fmaxf(part[tid], part[BM + tid]), fmaxf(part[2 * BM + tid], part[3 * BM + tid]));
                uint32_t *line = p.rq_rowmax + ((size_t)row0 + tid) * kRqCols;
                __hip_atomic_store(line + col_tile, __float_as_uint(m) | 0x80000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int ncols = p.N / BN;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)p.rq_rowmax, 0, 0x7FFFFFFF, 0x00020000);
                const uint32_t off = (uint32_t)(row0 + tid) * (kRqCols * 4u);
                const uint64_t t0 = ticks_100mhz();
                for (;;) {
                    asm volatile("" ::: "memory");             // (the line is re-read every round)
                    u32x4 w[kRqCols / 4];
#pragma unroll
                    for (int j = 0; j < kRqCols / 4; ++j)
                        w[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + 16u * j), 0, 16));     // sc1
                    uint32_t all = 0x80000000u, mx = 0u;
#pragma unroll
                    for (int j = 0; j < kRqCols; ++j) {
                        const uint32_t x = j < ncols ? w[j / 4][j % 4] : 0x80000000u;
                        all &= x;
                        mx = max(mx, x & 0x7FFFFFFFu);
                    }
                    if (all) {
                        amax = __uint_as_float(mx);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    if (ticks_100mhz() - t0 > p.timeout_ticks) {
                        report_status(p.status, MI_EP_STATUS_GEMM_ROWMAX);
                        break;
                    }
                }
                if (col_tile == 0) p.q_scale[(size_t)row0 + tid] = amax / 127.0f;
            }
            invs[tid] = amax > 0.f ? 1.0f / amax : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int lr = wm * kWaveRows + mt * 16 + c16;
            const float inv = invs[lr];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int a = (int)rintf((v[mt][nt][0] * 127.0f) * inv), b = (int)rintf((v[mt][nt][1] * 127.0f) * inv);
                const int c = (int)rintf((v[mt][nt][2] * 127.0f) * inv), d = (int)rintf((v[mt][nt][3] * 127.0f) * inv);
                *(uint32_t *)(qt + lr * kEpiRowBytes + f * 64 + h * 32 + nt * 16 + 4 * g) =
                    (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
            }
        }
        __syncthreads();
        // whole 128-byte pieces of the int8 rows: a wave-store covers 8 rows
        for (int it = wave; it < BM / 8; it += 16) {
            const int lr = it * 8 + (lane >> 3), chunk = lane & 7;
            if (lr < rows)
                *(u32x4 *)(p.q_out + ((size_t)row0 + lr) * (size_t)(p.N / 2) + (size_t)col_tile * 128 + chunk * 16) =
                    *(const u32x4 *)(qt + lr * kEpiRowBytes + chunk * 16);
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rl = mt * 16 + c16;
        const float as = p.a_scale[(size_t)row0 + min(wm * kWaveRows + rl, rows - 1)];
        if (MODE == 0) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gate = ((float)acc[mt][nt][r] * wsv[nt][r]) * as;
                    const float up = ((float)acc[mt][nt + 2][r] * wsv[nt + 2][r]) * as;
                    // sigmoid through v_rcp_f32 (1 ulp) instead of an IEEE division (~10 VALU operations per element): the fast exponential
                    // next to it is a few ulp off libm already, and the parity bar of this stage is rtol 3e-5 (tests/test_moe_gemm_gpu.py)
                    o[r] = up * (gate * __builtin_amdgcn_rcpf(1.0f + __expf(-gate)));
                }
                *(f32x4 *)(tile + rl * kRowBytes + (nt * 16 + 4 * g) * 4) = o;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                float d[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = ((float)acc[mt][nt][r] * wsv[nt][r]) * as;
                uint2 o;                                    // round to nearest even, as f32_to_bf16_rne (the products are finite)
                o.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{d[0], d[1]}, bf16x2));
                o.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{d[2], d[3]}, bf16x2));
                *(uint2 *)(tile + rl * kRowBytes + (nt * 16 + 4 * g) * 2) = o;
            }
        }
    }
    // a wave only reads back what it wrote itself: LDS operations of one wave complete in order, no barrier needed -- but the compiler
    // must not move the 16-byte loads below above the stores above (different types: type-based alias analysis would allow it; an
    // 8-rows-at-a-time variant of this epilogue produced wrong rows exactly that way)
    asm volatile("" ::: "memory");
    if (!wave_cols_ok) return;
    // MODE 2: where do this lane's rows go?  The (src, t, k) triples and, from them, the slot addresses are requested in ONE batch for all 8
    // row groups (after the transposition, when the accumulators are dead and the registers are free), two dependent round trips in all.
    // (Loaded inside the store loop, each of the 8 row groups paid both round trips in front of its store: GEMM2's epilogue cost
    // 18k cycles per tile against 11k for GEMM1, which writes as many bytes.)
    uint16_t *push_row[kWaveRows / 8];
    if (MODE == 2) {
        const int32_t *__restrict__ sidx = p.src_idx;
        int ts[kWaveRows / 8], tt[kWaveRows / 8], tk[kWaveRows / 8];
#pragma unroll
        for (int it = 0; it < kWaveRows / 8; ++it) {
            const int lr = wm * kWaveRows + it * 8 + (lane >> 3);
            const size_t grow = (size_t)row0 + min(lr, rows - 1);
            ts[it] = sidx[grow * 3 + 0], tt[it] = sidx[grow * 3 + 1], tk[it] = sidx[grow * 3 + 2];
        }
        const size_t poff = parity_off(p.par);
#pragma unroll
        for (int it = 0; it < kWaveRows / 8; ++it) {
            const int src = ts[it], t = tt[it], k = tk[it];
            // corrupted handle: drop instead of a wild (cross-GPU) store
            const bool ok = src >= 0 && src < p.W && k >= 0 && k < p.topk && t >= 0 && (long long)t * p.topk + k < p.slot_rows;
            push_row[it] = ok ? (uint16_t *)((uint8_t *)p.dsts.p[ok ? src : 0] + poff + ((size_t)t * p.topk + k) * p.slot_stride) : nullptr;
        }
    }
    // the rows are read back from the tile in one batch (always valid LDS reads), then stored: inside the conditional store loop every row
    // was an LDS round trip in front of its store
    u32x4 vrow[kWaveRows / 8];
#pragma unroll
    for (int it = 0; it < kWaveRows / 8; ++it) vrow[it] = *(const u32x4 *)(tile + (it * 8 + (lane >> 3)) * kRowBytes + (lane & 7) * 16);
#pragma unroll
    for (int it = 0; it < kWaveRows / 8; ++it) {
        const int rl = it * 8 + (lane >> 3), chunk = lane & 7;
        const int lr = wm * kWaveRows + rl;
        if (lr >= rows) continue;
        const size_t grow = (size_t)row0 + lr;
        const u32x4 v = vrow[it];
        if (MODE == 0) {
            float *orow = (float *)p.out + grow * (size_t)(p.N / 2) + (size_t)(col_tile * 2 + f) * 64 + h * 32;
            *(u32x4 *)(orow + chunk * 4) = v;
        } else {
            const int col = n0 + wn * 64 + chunk * 8;
            uint16_t *orow;
            if (MODE == 2) {
                if (!push_row[it]) continue;
                orow = push_row[it] + col;
            } else {
                orow = (uint16_t *)p.out + grow * (size_t)p.N + col;
            }
            if (col + 8 <= p.N) {
                *(u32x4 *)orow = v;
            } else {
                for (int j = 0; j < 8 && col + j < p.N; ++j) orow[j] = (uint16_t)(v[j >> 1] >> (16 * (j & 1)));
            }
        }
    }
#ifdef GEMM_TIMING
    const uint64_t t_issued = __builtin_amdgcn_s_memtime();    // all stores issued (not yet drained)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && blockIdx.x == 3 && blockIdx.y < 4) {
        g_gemm_dbg[(blockIdx.y * 16 + wave) * 4 + 2] = (float)(t_issued - t_epi);
        g_gemm_dbg[(blockIdx.y * 16 + wave) * 4 + 3] = (float)(__builtin_amdgcn_s_memtime() - t_epi);
    }
#endif
}

// The grid's y dimension is a POOL of tile workers, not one workgroup per possible tile: with worst-case sized buffers (fused_deep_moe
// at EP = 8 allocates W x max_tokens x K rows, 8x what arrives under balanced routing) one workgroup per possible tile would
// launch thousands that only look up "no such tile" and exit.  Worker y takes tile slots y, y + gridDim.y, ... until the
// cumulative counts say there are no more.
// Measured and dropped (round 2): running a worker's tiles as ONE operand stream -- the last refills of a tile fetch the first stages
// of the next one, the epilogue moves to its own 18 KB LDS area (8 rows per pass) so the ring keeps filling under it, the expert ends
// sit in LDS so the next lookup needs no vector load.  Bit-exact, but GEMM1 1.05 -> 1.10 ms and GEMM2 0.74 -> 0.76 ms at C5: only the
// kStages - 1 stages already requested overlap the epilogue (the accumulators occupy the registers a second tile would need, so the
// MFMA stream still stops for the ~11k-cycle epilogue), which buys back the pipeline fill (~5 % of a GEMM2 tile) and loses it again
// to the ring bookkeeping inside the k-loop.
// Measured and dropped (round 5): the same idea with the k-loop left alone -- between a tile's k-loop and its epilogue the worker looks up its
// NEXT tile and requests stage 0 of its ring (the epilogue's transposition tiles moved above the first stage, the 256-row tile through
// them in two passes); bit-exact, and no change at all: GEMM1 823.5 / 823.5 us, GEMM2 487.9 / 489.1, multinomial counts 895.9 / 901.2 and
// 530.4 / 533.7 (tools/time_gemm.py, one box).  The first k-tile's round trip is not what a tile waits for; like the MLA loop these
// kernels run at the chip's power limit (1.8-2.0 GHz, below), where re-timing a tile moves nothing.
// Shader clock under this kernel (always on: a handful of scalar instructions in ONE workgroup): the first workgroup stamps the shader-clock
// counter and the 100 MHz reference around its whole run.  mi_ep_moe_gemm_clock() turns the last launch's pair into GHz -- the chip
// does not hold its 2.4 GHz under dense INT8 MFMA issue, and the datasheet peak scales with the clock it does hold (bench.py reports the
// achieved rate against that effective peak beside the datasheet one).
__device__ unsigned long long g_gemm_clk[3][2];
template <int MODE, int MT, int BKT>
__global__ __launch_bounds__(kGemmThreads) void grouped_gemm_i8_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];    // kStages x (A BM x BKT + B 256 x BKT)
    const int lane0 = threadIdx.x & 63;
    const bool stamp = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    const unsigned long long c0 = stamp ? __builtin_amdgcn_s_memtime() : 0ull, r0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // Which column tile and which worker of the pool this workgroup is.  The hardware deals workgroups to the 8 XCDs by their flat id; a
    // column tile should always land on the same XCD (its weight tile then stays in that XCD's L2 for all row tiles of the expert).  With 16
    // column tiles (GEMM1) the plain (x, y) grid does that; with 28 (GEMM2: 7168 / 256) flat id x + 28 y puts column x on XCD (x + 4 y) % 8,
    // two XCDs per weight tile.  p.cols_padded = the column count rounded up to a multiple of 8: the flat id is cut by THAT, the surplus
    // columns leave at once.
    int col_tile = blockIdx.x, worker = blockIdx.y, workers = gridDim.y;
    if (p.cols_padded) {
        const int f = blockIdx.x + gridDim.x * blockIdx.y;
        col_tile = f % p.cols_padded, worker = f / p.cols_padded, workers = (int)(gridDim.x * gridDim.y) / p.cols_padded;
        if (col_tile * BN >= p.N || worker >= workers) return;
    }
#ifdef GEMM_SCRAMBLE      // measurement only: what column tile -> XCD consistency is worth (1: none at all; 2: a quarter of the workers off by 3)
    if (GEMM_SCRAMBLE == 1) col_tile = (col_tile + worker) % (int)gridDim.x;
    else if ((worker & 3) == 0) col_tile = (col_tile + 3) % (int)gridDim.x;
#endif
    if constexpr (MODE == 3) {
        // The column tiles of a row block wait for each other in the epilogue, so WHICH workgroups form a worker (one per column tile, walking
        // the same tile slots) is decided when they start, not by their block ids: the n-th workgroup to start on XCD x takes column tile
        // x + X (n mod c) of worker n / c (X XCDs, c = column tiles per XCD).  A waiter then only waits for workgroups that are running or
        // that the hardware starts next on a free CU of their XCD, whatever the dispatch order; and a column tile stays on one XCD (its
        // weight tile in that L2 for all row blocks of the expert: without that GEMM1 is 10 % slower, tools/time_gemm.py + -DGEMM_SCRAMBLE=1).
        // The one thing assumed is that the XCDs get equal shares of the grid (block b on XCD b mod X: mi_ep_moe_probe_xcds checks it at
        // start-up and rq_xcds = 1 -- one ticket for all, no placement assumed -- is used when it does not hold); every wait is bounded.
        __shared__ int s_col, s_worker;
        if (threadIdx.x == 0) {
            const int X = p.rq_xcds, cpx = (int)gridDim.x / X;
            const int xcc = X > 1 ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7) % X : 0;      // HW_REG_XCC_ID[3:0]
            const int n = (int)__hip_atomic_fetch_add(p.rq_tickets + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_col = xcc + X * (n % cpx), s_worker = n / cpx;
        }
        __syncthreads();
        col_tile = s_col, worker = s_worker;
        if (worker >= workers) return;                      // (more workgroups on this XCD than its share: nothing is left for them)
    }
    const int end_first = lane0 < p.L ? p.cum[(lane0 + 1) * p.cum_stride - 1] : 0;     // end of expert `lane` (cumulative row count)
    for (int slot = worker;; slot += workers) {
        int e, row0, rows;
        if (!find_tile<64 * MT>(p, slot, end_first, e, row0, rows)) break;
        // An expert's LAST row block is whatever is left of its rows.  Under multinomial routing about half the experts of a prefill batch
        // end in a block of a few dozen rows (1024 +- 30 rows per expert at C5: 4 full blocks and a fifth of <= 64 rows), and a 256-row
        // tile pays for it in full -- its weight tile streams all the same and the MFMAs multiply padding.  Blocks of <= 64 / <= 128 rows
        // run as the 64- / 128-row tile (a fifth / three eighths less operand stream per k-tile, a quarter / half of the MFMAs).  Same
        // products, same epilogue: bit-identical.
        if (MT == 4 && BKT == 128 && p.small_last && rows <= 64) gemm_tile<MODE, 1, BKT>(p, e, row0, rows, col_tile, lds);
        else if (MT == 4 && BKT == 128 && p.small_last && rows <= 128) gemm_tile<MODE, 2, BKT>(p, e, row0, rows, col_tile, lds);
        else gemm_tile<MODE, MT, BKT>(p, e, row0, rows, col_tile, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this tile's stores are out before the ring is refilled ...
        __syncthreads();                                         // ... and every wave is done with its epilogue tile in LDS
    }
    if (stamp) {
        g_gemm_clk[MODE % 3][0] = __builtin_amdgcn_s_memtime() - c0;      // (mode 3 is GEMM1 too)
        g_gemm_clk[MODE % 3][1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// per-row symmetric requantisation of the SwiGLU output: q = rint((v * 127) * (1 / rowmax)), scale = rowmax / 127
// (reference ...swiglu_quant_multistage_workspace.h:199-265).  One wave per row.
__global__ __launch_bounds__(256) void rowquant_kernel(const float *__restrict__ v, const int32_t *__restrict__ total_dev, int M_cap,
                                                      int I, int8_t *__restrict__ q, float *__restrict__ scale)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = min(*total_dev, M_cap);
    if (row >= total) return;
    const float *vr = v + row * (long long)I;
    float amax = 0.f;
    if (I <= 2048) {
        // the row stays in registers between the max pass and the quantisation (8 x 16 B per lane): v is read ONCE -- the PMC
        // counters showed 600 MB per launch against 336 MB algorithmic when the second pass re-read it
        float4 x[8];
        // unconditional loads (index clamped into the row; I % 4 == 0): conditional ones get a block and a vmcnt(0) each
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *(const float4 *)(vr + min(lane * 4 + u * 256, I - 4));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (lane * 4 + u * 256 >= I) x[u] = float4{0.f, 0.f, 0.f, 0.f};
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x[u].x), fabsf(x[u].y)), fmaxf(fabsf(x[u].z), fabsf(x[u].w))));
        }
        amax = wave_max(amax);
        const float inv = amax > 0.f ? 1.0f / amax : 0.f;
        if (lane == 0) scale[row] = amax / 127.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = lane * 4 + u * 256;
            if (i >= I) continue;
            const int a = (int)rintf((x[u].x * 127.0f) * inv), b = (int)rintf((x[u].y * 127.0f) * inv);
            const int c = (int)rintf((x[u].z * 127.0f) * inv), d = (int)rintf((x[u].w * 127.0f) * inv);
            *(uint32_t *)(q + row * (long long)I + i) =
                (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
        }
        return;
    }
    for (int i = lane * 4; i < I; i += 256) {
        const float4 x = *(const float4 *)(vr + i);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 1.0f / amax : 0.f;
    if (lane == 0) scale[row] = amax / 127.0f;
    for (int i = lane * 4; i < I; i += 256) {
        const float4 x = *(const float4 *)(vr + i);
        const int a = (int)rintf((x.x * 127.0f) * inv), b = (int)rintf((x.y * 127.0f) * inv);
        const int c = (int)rintf((x.z * 127.0f) * inv), d = (int)rintf((x.w * 127.0f) * inv);
        *(uint32_t *)(q + row * (long long)I + i) =
            (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
    }
}

}  // namespace mi_ep

using namespace mi_ep;

template <int MODE, int MT, int BKT>
static void gemm_launch_one(const GemmArgs &p, void *stream)
{
    constexpr int BM = 64 * MT;
    // operand ring | epilogue tiles of 16 waves; the 256-row kernel with 128-byte k-tiles also runs 64- and 128-row tiles (the last row
    // block of an expert): their rings are three stages deep, the 128-row one (3 x 48 KB) is the largest
    constexpr int ring0 = ring_bytes(BKT, MT), ring = (MT == 4 && BKT == 128 && ring_bytes(BKT, 2) > ring0) ? ring_bytes(BKT, 2) : ring0;
    constexpr int epi = MODE == 3 ? 8192 + 256 * kEpiRowBytes : 16 * 16 * MT * kEpiRowBytes;      // mode 3: maxima + the workgroup's int8 tile
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute((const void *)grouped_gemm_i8_kernel<MODE, MT, BKT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    const int tiles_max = (p.M_cap + BM - 1) / BM + p.L;        // every expert may end in a partial tile
    const int gx = (p.N + BN - 1) / BN;
    int pool = 2048 / gx > 8 ? 2048 / gx : 8;                   // ~8 workgroups per CU in flight over the launch; the rest is looped
    pool = tiles_max < pool ? tiles_max : pool;
    GemmArgs q = p;
    // Opt-in (MI_GEMM_XCD_COLS=1).  Measured at C5 shapes, round 4 (tools/probes/gemm_xcd_ab.sh, counters FETCH_SIZE x 2 + WRITE_SIZE):
    // GEMM2 moves 1.63 GB instead of 2.39 GB past its L2s (1.01 GB algorithmic) -- and takes 525 us instead of 482 us: 28 column tiles on 8
    // XCDs are 4 + 4 + 4 + 4 + 3 + 3 + 3 + 3, the four-tile XCDs finish 14 % later, and the plain order's (x + 4 y) % 8 IS the balanced deal
    // (every XCD 3.5 tiles on average, each weight tile on two XCDs).  The traffic past L2 is served by the 256 MB memory-side cache; it
    // is not what the kernel waits for.  GEMM1 (16 column tiles) is unaffected: 802 us either way, 3.09 GB against 1.44 GB algorithmic.
    // (A balanced variant -- the surplus column slots 28 .. 31, on XCDs 4 .. 7, taking every second (expert, row block) of columns 24 .. 27,
    //  3.5 columns per XCD -- was measured too: GEMM2 522 us uniform / 569 us with multinomial row counts against 482-490 / 537 for the plain
    //  order, fabric traffic 2.25 GB against 2.61 GB.  Consistency costs time even when the deal is even.)
    // MI_GEMM_SMALL_LAST: 0 never, 1 (default) both GEMMs, 2 GEMM1 only (A/B of GEMM2's fabric traffic: profiles/r06_gemm_traffic_ab.txt)
    static const int small_last = getenv("MI_GEMM_SMALL_LAST") ? atoi(getenv("MI_GEMM_SMALL_LAST")) : 1;
    q.small_last = small_last == 1 || (small_last == 2 && (MODE == 0 || MODE == 3));
    static const bool xcd_cols = getenv("MI_GEMM_XCD_COLS") && atoi(getenv("MI_GEMM_XCD_COLS")) != 0;
    q.cols_padded = (xcd_cols && gx % 8 != 0 && gx > 8) ? (gx + 7) / 8 * 8 : 0;
    // (padded form: the same number of workers, gx' x pool flat ids cut into rows of gx')
    dim3 grid(q.cols_padded ? q.cols_padded : gx, pool);
    grouped_gemm_i8_kernel<MODE, MT, BKT><<<grid, kGemmThreads, lds, (hipStream_t)stream>>>(q);
}

struct PushArgs {
    const int32_t *src_idx;
    PeerPtrs dsts;
    size_t slot_stride;
    int topk, W;
    Parity par;
    int slot_rows;
};

struct RequantArgs {
    int8_t *q;
    float *scale;
    uint32_t *words;          // mi_ep_moe_requant_words(rows_cap, L) zeroed uint32
    int xcds;
    int32_t *status;
    int timeout_ms;
};

static int gemm_launch(int mode, const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale, const int32_t *cum,
                       int cum_stride, int L, int M_cap, int K, int N, void *out, int rows_per_expert_hint, void *stream,
                       const PushArgs *push = nullptr, const uint32_t *a_rows = nullptr, const RequantArgs *rq = nullptr)
{
    if (!a || !a_scale || !w || !w_scale || !cum || (!out && !push && !rq) || L <= 0 || L > 1024 || M_cap <= 0 || K <= 0 || K % BK || N <= 0 ||
        N % 128 || cum_stride <= 0)
        return MI_EP_EINVAL;
    GemmArgs p{};
    p.a = a, p.a_scale = a_scale, p.w = w, p.w_scale = w_scale, p.cum = cum, p.cum_stride = cum_stride, p.L = L, p.M_cap = M_cap;
    p.K = K, p.N = N, p.out = out, p.a_rows = a_rows;
    const bool small = rows_per_expert_hint > 0 && rows_per_expert_hint <= 96;      // decode-size groups
    if (push) {
        p.src_idx = push->src_idx, p.dsts = push->dsts, p.slot_stride = push->slot_stride, p.topk = push->topk, p.W = push->W;
        p.par = push->par, p.slot_rows = push->slot_rows;
        mode = 2;
    }
    if (rq) {
        const int gx = N / BN;
        if (!rq->q || !rq->scale || !rq->words || !rq->status || N % BN || gx > kRqCols || rq->xcds < 1 || rq->xcds > 8) return MI_EP_EINVAL;
        // (fewer column tiles than XCDs, or not a multiple: one ticket for all workgroups)
        p.q_out = rq->q, p.q_scale = rq->scale, p.rq_xcds = gx % rq->xcds == 0 ? rq->xcds : 1, p.status = rq->status;
        p.rq_tickets = rq->words, p.rq_rowmax = rq->words + 16;      // (lines 64-byte aligned)
        p.timeout_ticks = (uint64_t)(rq->timeout_ms > 0 ? rq->timeout_ms : 10000) * 100000ull;
        mode = 3;
    }
    const bool wide_k = small && K % 128 == 0;             // decode tile with whole 128-byte lines per request
    static const bool wide_big_env = !(getenv("MI_GEMM_WIDE_K") && atoi(getenv("MI_GEMM_WIDE_K")) == 0);      // 0: 64-byte k-tiles (A/B runs)
    const bool wide_big = !small && wide_big_env && K % 128 == 0;
#define MI_GEMM_DISPATCH(M)                                                          \
    do {                                                                             \
        if (wide_k) gemm_launch_one<M, 1, 128>(p, stream);                           \
        else if (small) gemm_launch_one<M, 1, 64>(p, stream);                        \
        else if (wide_big) gemm_launch_one<M, 4, 128>(p, stream);                    \
        else gemm_launch_one<M, 4, 64>(p, stream);                                   \
    } while (0)
    if (mode == 0) MI_GEMM_DISPATCH(0);
    else if (mode == 1) MI_GEMM_DISPATCH(1);
    else if (mode == 3) MI_GEMM_DISPATCH(3);
    else MI_GEMM_DISPATCH(2);
#undef MI_GEMM_DISPATCH
    return launch_status();
}

extern "C" int mi_ep_moe_gemm1_swiglu(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale,
                                      const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap, int hidden,
                                      int two_i, float *out, int rows_per_expert_hint, void *stream)
{
    return gemm_launch(0, a, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, hidden, two_i, out,
                       rows_per_expert_hint, stream);
}

extern "C" int mi_ep_moe_gemm1_swiglu_rows(const void *a_base, const uint32_t *a_row_offsets, const float *a_scale, const int8_t *w,
                                           const float *w_scale, const int32_t *row_cumsum, int cum_stride, int num_local_experts,
                                           int rows_cap, int hidden, int two_i, float *out, int rows_per_expert_hint, void *stream)
{
    if (!a_row_offsets) return MI_EP_EINVAL;
    return gemm_launch(0, (const int8_t *)a_base, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, hidden, two_i,
                       out, rows_per_expert_hint, stream, nullptr, a_row_offsets);
}

extern "C" size_t mi_ep_moe_requant_words(int rows_cap, int num_local_experts)
{
    (void)num_local_experts;
    return (size_t)16 + (size_t)rows_cap * kRqCols;
}

extern "C" int mi_ep_moe_gemm1_swiglu_quant(const void *a_base, const uint32_t *a_row_offsets, const float *a_scale, const int8_t *w,
                                            const float *w_scale, const int32_t *row_cumsum, int cum_stride, int num_local_experts,
                                            int rows_cap, int hidden, int two_i, int8_t *q, float *q_scale, uint32_t *zeroed_words, int xcds,
                                            int32_t *status, int timeout_ms, int rows_per_expert_hint, void *stream)
{
    RequantArgs rq{q, q_scale, zeroed_words, xcds, status, timeout_ms};
    return gemm_launch(0, (const int8_t *)a_base, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, hidden, two_i,
                       nullptr, rows_per_expert_hint, stream, nullptr, a_row_offsets, &rq);
}

// Which XCD does block b of a grid run on?  64 one-wave blocks write their HW_REG_XCC_ID; the answer the requantising GEMM1 relies on for
// SPEED and for equal shares is "b mod X".  Returns X (8 on an MI355X in SPX mode, 1 on a single-XCD partition), or 1 -- no placement
// assumed -- when the observed ids do not follow that rule.
namespace mi_ep {
__global__ void probe_xcc_kernel(int32_t *out) { if (threadIdx.x == 0) out[blockIdx.x] = (int32_t)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15); }
}
extern "C" int mi_ep_moe_probe_xcds(void *stream)
{
    int32_t *dev = nullptr, host[64];
    if (hipMalloc((void **)&dev, sizeof(host)) != hipSuccess) return 1;
    int X = 1;
    mi_ep::probe_xcc_kernel<<<64, 64, 0, (hipStream_t)stream>>>(dev);
    if (hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, (hipStream_t)stream) == hipSuccess &&
        hipStreamSynchronize((hipStream_t)stream) == hipSuccess) {
        for (int cand = 8; cand > 1 && X == 1; cand >>= 1) {
            bool ok = true;
            for (int b = 0; b < 64 && ok; ++b) ok = host[b] == b % cand;
            if (ok) X = cand;
        }
    }
    (void)hipFree(dev);
    return X;
}

extern "C" int mi_ep_moe_rowquant(const float *v, const int32_t *total_rows_dev, int rows_cap, int inter, int8_t *q, float *scale,
                                  void *stream)
{
    if (!v || !total_rows_dev || !q || !scale || rows_cap <= 0 || inter <= 0 || inter % 4) return MI_EP_EINVAL;
    rowquant_kernel<<<(rows_cap + 3) / 4, 256, 0, (hipStream_t)stream>>>(v, total_rows_dev, rows_cap, inter, q, scale);
    return launch_status();
}

extern "C" int mi_ep_moe_gemm2(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale,
                               const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap, int inter,
                               int hidden, void *out_bf16, int rows_per_expert_hint, void *stream)
{
    return gemm_launch(1, a, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, inter, hidden, out_bf16,
                       rows_per_expert_hint, stream);
}

extern "C" int mi_ep_moe_gemm2_push(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale,
                                    const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap, int inter,
                                    int hidden, const int32_t *src_idx, int topk, void *const *dst_base_host, int num_ranks,
                                    size_t slot_region_bytes, const uint64_t *epoch_ctr, size_t parity_stride,
                                    int rows_per_expert_hint, void *stream)
{
    if (!src_idx || !dst_base_host || topk <= 0 || topk > MI_EP_MAX_TOPK || num_ranks <= 0 || num_ranks > MI_EP_MAX_RANKS)
        return MI_EP_EINVAL;
    PushArgs push{};
    push.src_idx = src_idx, push.slot_stride = mi_ep_combine_row_bytes(hidden), push.topk = topk, push.W = num_ranks;
    push.par = make_parity(epoch_ctr, 1, parity_stride);
    push.slot_rows = slot_region_bytes ? (int)(slot_region_bytes / push.slot_stride < 0x7FFFFFFF ? slot_region_bytes / push.slot_stride : 0x7FFFFFFF)
                                       : 0x7FFFFFFF;
    for (int i = 0; i < num_ranks; ++i) {
        if (!dst_base_host[i]) return MI_EP_EINVAL;
        push.dsts.p[i] = dst_base_host[i];
    }
    return gemm_launch(1, a, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, inter, hidden, nullptr,
                       rows_per_expert_hint, stream, &push);
}

extern "C" int mi_ep_moe_gemm_clock(double *ghz3, double *us3)
{
    unsigned long long h[3][2];
    if (!ghz3 || !us3 || hipMemcpyFromSymbol(h, HIP_SYMBOL(mi_ep::g_gemm_clk), sizeof(h)) != hipSuccess) return MI_EP_EINVAL;
    for (int m = 0; m < 3; ++m) {
        us3[m] = (double)h[m][1] / 100.0;
        ghz3[m] = h[m][1] ? (double)h[m][0] / ((double)h[m][1] * 10.0) : 0.0;      // ticks per 10 ns -> GHz
    }
    return MI_EP_OK;
}

#ifdef GEMM_TIMING
extern "C" int mi_ep_gemm_dbg(float *host128)
{
    return (int)hipMemcpyFromSymbol(host128, HIP_SYMBOL(mi_ep::g_gemm_dbg), 256 * sizeof(float));
}
#endif
