// Length-aware work lists for the paged decode kernels (MLA: mla_decode*.hip; GQA: gqa_decode.hip), built on the device from kv_seq_lens.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi_sgl {

// ---- the planned form ---------------------------------------------------------------------------------------------------------------
// A launch with per-sequence lengths far apart (a serving batch) runs at the pace of its longest sequence when every sequence is cut
// into the same number of splits.  The plan cuts each (sequence, kv head) into n_s = ceil(tiles_s / x) pieces, x = the smallest piece
// size for which all pieces together are no more than the chip runs at once (one workgroup per CU), and orders the pieces longest
// first.  One workgroup per piece ("item"); the partial of item i lives at slot i of the workspace.
//   words [0, kPlanHdr)                 n_items (padding included), n_rounds, base[k] for k < kPlanMaxSplits
//   words [kPlanHdr, + 2 seqs)          per sequence: rank (position by descending cost), n_s
//   words [.., + 4 items_max)           per item: seq (-1 = padding), first tile, end tile, k | n_s << 8
// Item index of piece k of the sequence ranked r: base[k] + r.  n_s never increases with the rank, so the sequences with more than
// k pieces are exactly the ranks below cnt_k; rounds are stored highest k first (the pieces of the longest sequences lead, the short
// unsplit sequences come last: longest-processing-time-first for the hardware's in-order dispatch) and every base[k] is a multiple of
// 8: all pieces of a sequence have the same index mod 8, i.e. run on one XCD (dispatch convention), where Q^T and the partials meet in L2.
constexpr int kPlanHdr = 32, kPlanMaxSplits = 16, kPlanMinTiles = 8, kPlanSortMax = 2048;
__host__ __device__ inline long long plan_items_max(long long seqs, int workers) { return seqs + workers + 8 * kPlanMaxSplits; }
__host__ __device__ inline size_t plan_words(long long seqs, int workers)
{
    return (size_t)kPlanHdr + 2 * (size_t)seqs + 4 * (size_t)plan_items_max(seqs, workers);
}
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
#define PLAN_T(i) if (threadIdx.x == 0) ((volatile int32_t *)plan)[20 + (i)] = (int32_t)(__builtin_amdgcn_s_memrealtime() & 0x7FFFFFFF);
#else
#define PLAN_T(i)
#endif
static __global__ __launch_bounds__(1024) void decode_plan_kernel(const int32_t *__restrict__ seq_lens, int batch, int kv_heads, int tile, int align,
                                                                  int workers, int32_t *__restrict__ plan)
{
    __shared__ int s_tiles[kPlanSortMax], s_rn[kPlanSortMax];      // tiles; rank | n << 16 (kept for the last pass)
    __shared__ int s_cnt[kPlanMaxSplits], s_base[kPlanMaxSplits], s_cand[16];
    __shared__ long long s_total;
    __shared__ int s_max, s_lo, s_hi;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, seqs = batch * kv_heads;
    const long long items_max = plan_items_max(seqs, workers);
    int32_t *info = plan + kPlanHdr, *items = plan + kPlanHdr + 2ll * seqs;
    PLAN_T(0)
    if (tid == 0) s_total = 0, s_max = 0;
    if (tid < kPlanMaxSplits) s_cnt[tid] = 0;
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
    for (long long i = tid; i < items_max; i += blockDim.x) items[4 * i] = -1;      // every slot starts as padding
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
    if (sorted) {
        // rank of a sequence = how many are longer (ties: lower index first).  Thread (s, part): the workgroup's threads are dealt over the
        // sequences, `parts` threads per sequence, each comparing a slice of the others; the partial counts meet in LDS.  (One thread per
        // sequence walking all the others, or one wave per sequence with ballots, took 3.7 us of the kernel's 8-9: serial LDS round trips.)
        int pad = 1;
        while (pad < seqs) pad <<= 1;
        const int parts = max(1, (int)blockDim.x / pad);          // power of two
        for (int s = tid; s < seqs; s += blockDim.x) s_rn[s] = 0;
        __syncthreads();
        for (int i = tid; i < pad * parts; i += blockDim.x) {      // (one pass when the batch has <= 1024 sequences; parts = 1 beyond)
            const int s = i & (pad - 1), part = i / pad;
            if (s < seqs) {
                const int t = s_tiles[s], per = (seqs + parts - 1) / parts;
                int c = 0;
                for (int o = part * per; o < min(seqs, (part + 1) * per); ++o) c += (s_tiles[o] > t) || (s_tiles[o] == t && o < s);
                if (c) atomicAdd(&s_rn[s], c);
            }
        }
        __syncthreads();
        for (int s = tid; s < seqs; s += blockDim.x) {
            const int rank = s_rn[s], n = split ? plan_pieces(s_tiles[s], x) : 1;
            info[2 * s] = rank, info[2 * s + 1] = n;
            s_rn[s] = rank | (n << 16);
            for (int k = 0; k < n; ++k) atomicAdd(&s_cnt[k], 1);
        }
    } else {
        for (int s = tid; s < seqs; s += blockDim.x) info[2 * s] = s, info[2 * s + 1] = 1;
    }
    PLAN_T(3)
    if (!sorted && tid == 0) s_cnt[0] = seqs;
    __syncthreads();
    if (tid == 0) {
        int rounds = 0, at = 0;
        for (int k = 0; k < kPlanMaxSplits; ++k) rounds += s_cnt[k] > 0;
        for (int k = kPlanMaxSplits - 1; k >= 0; --k) {                             // highest k first, every base a multiple of 8
            s_base[k] = at;
            at += (s_cnt[k] + 7) & ~7;
        }
        plan[0] = at, plan[1] = rounds;
        for (int k = 0; k < kPlanMaxSplits; ++k) plan[2 + k] = s_base[k];
    }
    __syncthreads();
    for (int s = tid; s < seqs; s += blockDim.x) {
        const int tiles = sorted ? s_tiles[s] : (max(seq_lens[s / kv_heads], 0) + tile - 1) / tile;      // (no second trip to global memory)
        const int rank = sorted ? (s_rn[s] & 0xFFFF) : s, n = sorted ? (s_rn[s] >> 16) : 1;
        const int per = ((tiles + n - 1) / n + align - 1) / align * align;      // (trailing pieces may come out empty: a partial of weight 0)
        for (int k = 0; k < n; ++k) {
            int32_t *it = items + 4ll * (s_base[k] + rank);
            it[1] = min(tiles, k * per), it[2] = min(tiles, (k + 1) * per), it[3] = k | (n << 8);
            it[0] = s;
        }
    }
    PLAN_T(4)
}


}  // namespace mi_sgl
