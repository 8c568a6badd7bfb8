// Length-aware work lists for the paged decode kernels (MLA: mla_decode*.hip; GQA: gqa_decode.hip), built on the device from kv_seq_lens.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi_sgl {

// ---- the planned form ---------------------------------------------------------------------------------------------------------------
// A launch with per-sequence lengths far apart (a serving batch) runs at the pace of its longest sequence when every sequence is cut
// into the same number of splits.  The plan cuts each (sequence, kv head) into n_s = ceil(tiles_s / x) pieces, x = the smallest piece
// size for which all pieces together are no more than the chip runs at once (one workgroup per CU), and lists the pieces sequence by
// sequence, longest sequence first.  One workgroup per piece ("item"); the partial of item i lives at slot i of the workspace.
//   words [0, kPlanHdr)                 n_items, piece size x (tiles)
//   words [kPlanHdr, + 2 seqs)          per sequence: first item, n_s
//   words [.., + 4 items_max)           per item: seq (-1 = behind the list), first tile, end tile, k | n_s << 8
// Item of piece k of sequence s: first[s] + k.  The pieces of a sequence are CONSECUTIVE items: the hardware deals workgroup i to XCD
// i mod 8, so any run of items spreads evenly over the XCDs, and "all pieces in one round of workgroups" holds per XCD as well.  (The first
// layout kept the pieces of a sequence on one XCD -- rounds of equal piece index, every round padded to a multiple of 8 -- so that Q^T
// and the partials of a sequence met in one L2.  It cost more than it gave: a sequence cut into 64 pieces ran on the 32 CUs of ONE XCD
// (batch 4 x 32k keys, 16 heads: 88 us against 49 for 64 uniform splits); with the pieces walking over the XCDs the piece COUNTS per XCD
// still differed by up to n_longest - n_shortest, and an XCD with 35 of 256 pieces runs three of them in a second round (batch 64 x
// U[1, 8192] keys: 152 us against 133 for four uniform splits, although the longest piece was 40 tiles against 64); and the padding --
// up to 7 items per round, 64 rounds -- put two empty workgroups in front of every real one for batches of a few long sequences.)
// Longest sequence first: longest-processing-time-first for the hardware's in-order dispatch, for the lists that are not cut at all.
// (64 pieces at most: a handful of very long sequences must still be able to fill the chip, as the uniform form's cap of 64 splits lets them.)
constexpr int kPlanHdr = 16, kPlanMaxSplits = 64, kPlanMinTiles = 8, kPlanSortMax = 2048;
__host__ __device__ inline long long plan_items_max(long long seqs, int workers) { return (seqs + workers + 7) / 8 * 8; }
__host__ __device__ inline size_t plan_words(long long seqs, int workers)
{
    return (size_t)kPlanHdr + 2 * (size_t)seqs + 4 * (size_t)plan_items_max(seqs, workers);
}
__device__ __forceinline__ int plan_item_index(const int32_t *plan, int seq, int k) { return plan[kPlanHdr + 2ll * seq] + k; }
struct PlanItem {
    int seq, t_begin, t_end, k, n;
};
__device__ __forceinline__ PlanItem plan_item(const int32_t *plan, long long seqs, int item)      // wave-uniform
{
    const int32_t *it = plan + kPlanHdr + 2 * seqs + 4ll * item;
    PlanItem r;
    r.seq = __builtin_amdgcn_readfirstlane(it[0]);
    r.t_begin = __builtin_amdgcn_readfirstlane(it[1]);
    r.t_end = __builtin_amdgcn_readfirstlane(it[2]);
    const int kn = __builtin_amdgcn_readfirstlane(it[3]);
    r.k = kn & 0xFF, r.n = kn >> 8;
    return r;
}

// The work list of the planned form (layout and rationale above).  One workgroup of 1024 threads; `tile` = keys per tile of the
// kernel that will consume the list, `workers` = workgroups the chip runs at once (one per CU for the wide kernels); pieces start on
// multiples of `align` tiles (MLA: 2 -- the list is in 32-key tiles and also serves the 64-head kernel, whose tiles are 64 keys).
// Piece size: the smallest x (tiles per piece) for which sum_s n_s(x) <= workers, n_s(x) = ceil(tiles_s / x) capped so that a piece
// keeps >= kPlanMinTiles tiles and a sequence <= kPlanMaxSplits pieces -- every piece then runs in the FIRST round of workgroups (a
// list of more items than CUs leaves the last items to a second round: measured, a ragged C4 batch cut by "cost / average cost" made
// 270 items and ran no faster than two uniform splits) and the longest piece is as short as that allows.  Found by a 16-way search, one
// candidate per wave (the range shrinks sixteen-fold per round).
__device__ __forceinline__ int plan_pieces(int tiles, int x)
{
    const int n = (tiles + x - 1) / x;
    return max(1, min(n, min(kPlanMaxSplits, tiles / kPlanMinTiles)));
}
#ifdef PLAN_TIMING
#define PLAN_T(i) if (threadIdx.x == 0) ((volatile int32_t *)plan)[8 + (i)] = (int32_t)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFF);
#else
#define PLAN_T(i)
#endif
static __global__ __launch_bounds__(1024) void decode_plan_kernel(const int32_t *__restrict__ seq_lens, int batch, int kv_heads, int tile, int align,
                                                                  int workers, int32_t *__restrict__ plan)
{
    __shared__ int s_tiles[kPlanSortMax], s_rank[kPlanSortMax], s_first[kPlanSortMax];      // tiles; rank; pieces by rank, then their exclusive sum
    __shared__ int s_cand[16], s_wsum[16];
    __shared__ long long s_total;
    __shared__ int s_max, s_lo, s_hi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, seqs = batch * kv_heads;
    const long long items_max = plan_items_max(seqs, workers);
    int32_t *info = plan + kPlanHdr, *items = plan + kPlanHdr + 2ll * seqs;
    PLAN_T(0)
    if (tid == 0) s_total = 0, s_max = 0;
    __syncthreads();
    // batches of more sequences than the sort handles, or than the chip has CUs, run unsplit (one piece each: they fill the chip as they are)
    const bool sorted = seqs <= kPlanSortMax;
    const bool split = sorted && seqs < workers;
    long long mine = 0;
    int mx = 0;
    for (int s = tid; s < seqs; s += blockDim.x) {
        const int t = (max(seq_lens[s / kv_heads], 0) + tile - 1) / tile;
        if (sorted) s_tiles[s] = t;
        mine += t, mx = max(mx, t);
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64), mx = max(mx, __shfl_xor(mx, o, 64));
    if (lane == 0 && mine) atomicAdd((unsigned long long *)&s_total, (unsigned long long)mine), atomicMax(&s_max, mx);
    for (long long i = tid; i < items_max; i += blockDim.x) items[4 * i] = -1;      // every slot starts as "behind the list"
    __syncthreads();
    PLAN_T(1)
    int x = max(s_max, 1);                                                           // one piece per sequence
    if (split) {
        if (tid == 0) s_lo = max(1, (int)((s_total + workers - 1) / workers)), s_hi = max(s_max, 1);
        __syncthreads();
        while (s_lo < s_hi) {                                                        // (LDS values: the same for every thread)
            const int lo = s_lo, hi = s_hi;
            const int step = max(1, (hi - lo + 15) / 16);
            const int xw = min(hi, lo + wave * step);                                // wave w's candidate; non-decreasing in w
            int cnt = 0;
            for (int s = lane; s < seqs; s += 64) cnt += plan_pieces(s_tiles[s], xw);
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            if (lane == 0) s_cand[wave] = cnt;
            __syncthreads();
            if (tid == 0) {
                int w = 0;
                while (w < 16 && s_cand[w] > workers) ++w;                           // first feasible candidate; hi itself always is
                s_hi = w < 16 ? min(hi, lo + w * step) : hi;
                s_lo = w > 0 ? min(s_hi, min(hi, lo + (w - 1) * step) + 1) : lo;     // the candidate before it is not
            }
            __syncthreads();
        }
        x = s_hi;
    }
    PLAN_T(2)
    if (!sorted) {
        // more sequences than the sort takes: the list is the batch in its own order, one piece each
        for (int s = tid; s < seqs; s += blockDim.x) {
            const int tiles = (max(seq_lens[s / kv_heads], 0) + tile - 1) / tile;
            info[2 * s] = s, info[2 * s + 1] = 1;
            int32_t *it = items + 4ll * s;
            it[1] = 0, it[2] = tiles, it[3] = 0 | (1 << 8);
            it[0] = s;
        }
        if (tid == 0) plan[0] = seqs, plan[1] = x;
        return;
    }
    // rank of a sequence = how many are longer (ties: lower index first).  Thread (s, part): the workgroup's threads are dealt over the
    // sequences, `parts` threads per sequence, each comparing a slice of the others; the partial counts meet in LDS.  (One thread per
    // sequence walking all the others, or one wave per sequence with ballots, took 3.7 us of the kernel's 8-9: serial LDS round trips.)
    {
        int pad = 1;
        while (pad < seqs) pad <<= 1;
        const int parts = max(1, (int)blockDim.x / pad);          // power of two
        for (int s = tid; s < kPlanSortMax; s += blockDim.x) s_rank[s] = 0, s_first[s] = 0;
        __syncthreads();
        for (int i = tid; i < pad * parts; i += blockDim.x) {      // (one pass when the batch has <= 1024 sequences; parts = 1 beyond)
            const int s = i & (pad - 1), part = i / pad;
            if (s < seqs) {
                const int t = s_tiles[s], per = (seqs + parts - 1) / parts;
                int c = 0;
                for (int o = part * per; o < min(seqs, (part + 1) * per); ++o) c += (s_tiles[o] > t) || (s_tiles[o] == t && o < s);
                if (c) atomicAdd(&s_rank[s], c);
            }
        }
        __syncthreads();
    }
    // pieces per sequence, filed by rank; their exclusive prefix sum = the first item of the sequence of that rank
    for (int s = tid; s < seqs; s += blockDim.x) s_first[s_rank[s]] = split ? plan_pieces(s_tiles[s], x) : 1;      // (ranks are a permutation)
    __syncthreads();
    PLAN_T(3)
    {
        // block-wide exclusive scan of s_first[0 .. 2048): two consecutive entries per thread, wave scans, 16 wave totals
        const int a = s_first[2 * tid], b = s_first[2 * tid + 1];
        int incl = a + b;
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_wsum[w];
        const int excl = off + incl - (a + b);
        s_first[2 * tid] = excl, s_first[2 * tid + 1] = excl + a;
        if (tid == 1023) plan[0] = excl + a + b, plan[1] = x;
    }
    __syncthreads();
    for (int s = tid; s < seqs; s += blockDim.x) {
        const int tiles = s_tiles[s], n = split ? plan_pieces(tiles, x) : 1, first = s_first[s_rank[s]];
        info[2 * s] = first, info[2 * s + 1] = n;
        const int per = ((tiles + n - 1) / n + align - 1) / align * align;      // (trailing pieces may come out empty: a partial of weight 0)
        for (int k = 0; k < n; ++k) {
            int32_t *it = items + 4ll * (first + k);
            it[1] = min(tiles, k * per), it[2] = min(tiles, (k + 1) * per), it[3] = k | (n << 8);
            it[0] = s;
        }
    }
    PLAN_T(4)
}


}  // namespace mi_sgl
