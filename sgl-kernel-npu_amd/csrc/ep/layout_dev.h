// Device-side pieces of the dispatch layout shared by layout.hip (the stand-alone launches) and dispatch.hip (the low-latency send
// kernel, whose first workgroup computes the layout while the others quantise): see layout.hip for the design.
#pragma once
#include "ep_common.h"

namespace mi_ep {

constexpr int kLayoutUnitTokens = 64;

template <bool I32>
__device__ __forceinline__ long long load_idx(const void *p, long long i)
{
    if (I32) return (long long)((const int32_t *)p)[i];
    return ((const long long *)p)[i];
}

// n / d for 0 <= n < 2048 * 64 and 1 <= d <= 2048 through a float reciprocal (inv = 1.0f / d): (n + 0.5) / d is at least 0.5 / d away
// from an integer, three orders of magnitude more than the rounding of the two float operations, so the truncation is exact.  An
// integer division is ~25 instructions, five of them quarter-rate; the histogram kernel had two per batch of 64 pairs and was
// VALU-bound after its loads were batched.
__device__ __forceinline__ int div_small(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

// lanes holding the same key as this lane (key < 2^nbits), via nbits ballots
__device__ __forceinline__ unsigned long long match_any_bits(unsigned key, bool active, int nbits)
{
    unsigned long long m = __ballot(active);
    for (int b = 0; b < nbits; ++b) {
        unsigned long long s = __ballot(active && ((key >> b) & 1u));
        m &= ((key >> b) & 1u) ? s : ~s;
    }
    return m;
}

// The three passes in ONE launch: workgroups of 16 waves, a wave per unit (16 units = 1024 tokens per workgroup with 64-token units;
// 16-token units for <= 256 tokens so that a 128-token decode batch occupies 8 waves instead of 2); the per-unit histograms never
// leave LDS.  Same arithmetic and the same deterministic slot order as the three kernels above.
//   one workgroup  (T <= 1024: decode / low-latency mode): removes two launches (~10 us of a ~60 us low-latency dispatch);
//   B workgroups   (larger batches, `sync` != NULL): every workgroup publishes the histogram of ITS 16 units (E + W words), all meet at
//     a grid barrier (B <= 128 co-resident workgroups; two self-resetting words in caller-owned zero-initialised memory), and each
//     derives the running base of its units from the totals of the workgroups in front of it: three launches (4.9 + 6.2 + 6.9 us
//     back to back at 4096 tokens) become one, and the [U][E] histograms / bases never travel through global memory.
// -> true when every workgroup arrived; false after a 2 s spin (a workgroup that never became resident) or when the arrival count is
// seen ABOVE the grid size (sync words shared with another launch in flight): the caller must then NOT trust the other workgroups' totals -- it reports MI_EP_STATUS_LAYOUT_BARRIER
// through `status` (the host's check_status raises) and poisons its outputs instead of writing plausible garbage.
// (ok_s: one word of the caller's dynamic LDS -- a static __shared__ word here would add to the 160 KB the launcher asks for)
__device__ __forceinline__ bool layout_grid_barrier(uint32_t *sync, int B, int32_t *status, int32_t *ok_s)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t0 = wall_clock64();
        bool ok = true;
        // relaxed polls, ONE acquire fence after the last arrival: an acquire load invalidates the XCD's L2 on every iteration, for every
        // workgroup of the XCD (tools/probes/ubench/launch_floor.hip: a flag hop between XCDs costs 0.4-0.7 us polled this way)
        uint32_t seen;
        while ((seen = __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < (uint32_t)B) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull) { ok = false; break; }      // 2 s: never hang
        }
        // more arrivals than this grid has workgroups: the pair of words is shared with another launch in flight (or was not zero when it
        // was lent) -- some workgroups left before every total was published.  Same report as a timeout.
        if (seen > (uint32_t)B) ok = false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (!ok && status) report_status(status, kStatusLayoutBarrier);
        // the last workgroup to LEAVE the spin re-arms both words for the next launch that borrows this pair
        if (__hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)B - 1u) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        *ok_s = ok ? 1 : 0;
    }
    __syncthreads();
    return *ok_s != 0;
}

// (a device function: the low-latency send kernel of dispatch.hip runs it in its first workgroup, B = 1, while the other workgroups
//  already quantise their rows)
template <bool I32, int UT>
__device__ __forceinline__ void layout_small_body(
    const void *__restrict__ topk_idx, int T, int K, int E, int W, int nbits, int32_t *__restrict__ num_tokens_per_rank,
    int32_t *__restrict__ num_tokens_per_expert, int32_t *__restrict__ is_token_in_rank,
    int32_t *__restrict__ send_token_idx_small, int32_t *__restrict__ send_data_offset,
    int32_t *__restrict__ block_tot /*[B][E + W], B > 1 only*/, uint32_t *__restrict__ sync, int32_t *smem, const int B, const int blk,
    int32_t *__restrict__ status = nullptr)
{
    const int U_all = (T + UT - 1) / UT;
    const int u_first = blk * 16;
    const int U = min(16, U_all - u_first);                                 // units of this workgroup (wave w owns unit u_first + w)
    int32_t *hist = smem;                                                   // [16][E], becomes the running base in pass 2
    unsigned long long *rmask = (unsigned long long *)(smem + 16 * E);      // [16][UT]
    int32_t *rank_cnt = (int32_t *)(rmask + 16 * UT);                       // [W]
    int32_t *wave_tot = rank_cnt + W;                                       // [16]
    int32_t *carry = wave_tot + 16;                                         // [1]
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / kWave;
#ifdef LAYOUT_TIMING
    uint64_t tk[8]; int ntk = 0;
#define LT_TICK() tk[ntk++] = wall_clock64();
#else
#define LT_TICK()
#endif
    LT_TICK()
    const int L = E / W;
    const float inv_k = 1.0f / (float)K, inv_l = 1.0f / (float)L;
    // ---- pass 1: histogram + token -> rank masks.  The unit's expert ids are requested in ONE batch and stay in registers for pass 3:
    // read batch by batch in both passes they were four to eight dependent global round trips of a ~6 us kernel.  (Requested before the
    // LDS tables are cleared: the clearing runs under the loads' latency.)
    constexpr int kB = UT * MI_EP_MAX_TOPK / kWave;               // batches of 64 (token, k) pairs in a unit
    long long ev[kB];
    const int unit = u_first + wave;                              // global unit of this wave (valid when wave < U)
    {
        const int unit0 = wave < U ? unit : u_first;
        const long long q0 = (long long)unit0 * UT * K;
        const int np0 = min(UT, T - unit0 * UT) * K;
#pragma unroll
        for (int i = 0; i < kB; ++i) ev[i] = (U > 0 && np0 > 0) ? load_idx<I32>(topk_idx, q0 + min(i * kWave + lane, np0 - 1)) : -1;
    }
    for (int i = tid; i < 16 * E; i += blockDim.x) hist[i] = 0;
    for (int i = tid; i < 16 * UT; i += blockDim.x) rmask[i] = 0ull;
    for (int i = tid; i < W; i += blockDim.x) rank_cnt[i] = 0;
    if (tid == 0) carry[0] = 0;
    __syncthreads();
    if (wave < U) {
        const int t0 = unit * UT;
        const int ntok = min(UT, T - t0);
        const int npairs = ntok * K;
        int32_t *h = hist + wave * E;
        unsigned long long *rm = rmask + wave * UT;
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            const int p = i * kWave + lane;
            if (i * kWave < npairs && p < npairs) {
                const long long e = ev[i];
                if (e >= 0 && e < E) {
                    atomicAdd(&h[(int)e], 1);
                    atomicOr(&rm[div_small(p, inv_k)], 1ull << div_small((int)e, inv_l));
                }
            }
        }
        const unsigned long long m = (lane < ntok) ? rm[lane] : 0ull;
        if (lane < ntok) {
            int32_t *row = is_token_in_rank + (long long)(t0 + lane) * W;
            for (int r = 0; r < W; ++r) row[r] = (int32_t)((m >> r) & 1ull);
        }
        for (int r = 0; r < W; ++r) {
            const unsigned long long b = __ballot((m >> r) & 1ull);
            if (lane == 0 && b) atomicAdd(&rank_cnt[r], __popcll(b));
        }
    }
    __syncthreads();
    LT_TICK()
    if (B > 1) {
        // this workgroup's totals -> global, everybody meets, then the totals of the workgroups in front of this one
        int32_t *mine = block_tot + (size_t)blk * (E + W);
        for (int e = tid; e < E; e += blockDim.x) {
            int32_t s = 0;
#pragma unroll 16
            for (int w = 0; w < 16; ++w) s += hist[w * E + e];
            mine[e] = s;
        }
        for (int r = tid; r < W; r += blockDim.x) mine[E + r] = rank_cnt[r];
        if (!layout_grid_barrier(sync, B, status, carry + 1)) {
            // barrier timed out: the tables of this launch cannot be formed.  The host learns through `status` (that is the contract);
            // the count tables are additionally set to -1 so that a caller who ignores it does not read plausible numbers.
            if (blk == B - 1) {
                for (int e = tid; e < E; e += blockDim.x) { num_tokens_per_expert[e] = -1; send_data_offset[e] = -1; }
                for (int r = tid; r < W; r += blockDim.x) num_tokens_per_rank[r] = -1;
            }
            return;
        }
    }
    LT_TICK()
    // ---- pass 2: per-expert exclusive scan over units (in place), totals, exclusive scan over experts.  With several workgroups a unit's
    // base starts at the sum of the earlier workgroups' totals; the LAST workgroup then holds the grand totals and writes the outputs.
    const bool writer = blk == B - 1;
    for (int base = 0; base < E; base += blockDim.x) {
        const int e = base + tid;
        int32_t run = 0;
        if (e < E) {
            for (int b0 = 0; b0 < blk; b0 += 16) {                   // batches of independent loads
                int32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = (b0 + j < blk) ? block_tot[(size_t)(b0 + j) * (E + W) + e] : 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) run += v[j];
            }
            for (int u = 0; u < 16; ++u) {
                const int32_t v = hist[u * E + e];
                hist[u * E + e] = run;
                run += v;
            }
            if (writer) num_tokens_per_expert[e] = run;
        }
        if (!writer) continue;                                     // workgroup-uniform
        const int32_t inc = wave_incl_scan_i32(run);
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        int32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        const int32_t c = carry[0];
        if (e < E) send_data_offset[e] = c + wbase + inc - run;
        __syncthreads();
        if (tid == blockDim.x - 1) carry[0] = c + wbase + inc;
        __syncthreads();
    }
    if (writer)
        for (int r = tid; r < W; r += blockDim.x) {
            int32_t s = rank_cnt[r];
            for (int b0 = 0; b0 < blk; ++b0) s += block_tot[(size_t)b0 * (E + W) + E + r];
            num_tokens_per_rank[r] = s;
        }
    // ---- pass 3: slot of every (t, k) inside its expert's segment
    __syncthreads();
    LT_TICK()
    if (wave < U) {
        const int t0 = unit * UT;
        const long long p0 = (long long)t0 * K;
        const int npairs = min(UT, T - t0) * K;
        int32_t *cnt = hist + wave * E;
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < kB; ++i) {
            const int c = i * kWave;
            if (c >= npairs) break;                                // wave-uniform
            const int p = c + lane;
            const long long e = p < npairs ? ev[i] : -1;           // the ids of pass 1
            const bool valid = (e >= 0 && e < E);
            const unsigned long long same = match_any_bits(valid ? (unsigned)e : 0u, valid, nbits);
            int32_t out = 0;
            if (valid) {
                const int before = __popcll(same & lt);
                const int32_t b0 = cnt[(int)e];
                out = b0 + before;
                if ((same >> lane) == 1ull) cnt[(int)e] = b0 + before + 1;
            }
            if (p < npairs) send_token_idx_small[p0 + p] = out;
        }
    }
#ifdef LAYOUT_TIMING
    LT_TICK()
    if (tid == 0 && blk == 0) for (int i = 0; i < ntk; ++i) ((uint64_t *)((char *)block_tot + (512 << 10)))[i] = tk[i];
#endif
}


template <bool I32, int UT>
__global__ __launch_bounds__(1024) void layout_small_kernel(
    const void *__restrict__ topk_idx, int T, int K, int E, int W, int nbits, int32_t *__restrict__ num_tokens_per_rank,
    int32_t *__restrict__ num_tokens_per_expert, int32_t *__restrict__ is_token_in_rank,
    int32_t *__restrict__ send_token_idx_small, int32_t *__restrict__ send_data_offset,
    int32_t *__restrict__ block_tot /*[B][E + W], B > 1 only*/, uint32_t *__restrict__ sync, int32_t *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    layout_small_body<I32, UT>(topk_idx, T, K, E, W, nbits, num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank,
                               send_token_idx_small, send_data_offset, block_tot, sync, smem, (int)gridDim.x, (int)blockIdx.x, status);
}

// bytes of dynamic LDS layout_small_body needs (16 units per workgroup)
inline size_t layout_small_lds_bytes(int E, int W, int unit_tokens) { return (size_t)16 * E * 4 + (size_t)16 * unit_tokens * 8 + (size_t)(W + 16 + 4) * 4; }

}  // namespace mi_ep
