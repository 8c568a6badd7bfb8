// A1 dispatch layout for gfx950.
// Replaces aclnnDispatchLayout (reference kernel csrc/deepep/ops/op_kernel/dispatch_layout.h:81-219).
//
// MI355X design: the (token,k) pairs are cut into units of 64 tokens, one wave64 per unit.
//   pass 1  per-unit expert histogram in LDS (LDS atomics) + token->rank bitmask,
//   pass 2  one workgroup: per-expert exclusive scan over units (coalesced over experts),
//           totals, exclusive scan over experts (send_data_offset), per-rank token counts,
//   pass 3  per unit, pairs are walked in row-major order 64 at a time; the rank of a pair among
//           equal experts inside the 64-wide step comes from ballots (no serial loop), the running
//           base lives in LDS.  Results are order-deterministic (no global atomics).
#include "ep_common.h"

namespace mi_ep {

constexpr int kUnitTokens = 64;
constexpr int kWavesPerBlock = 4;

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

constexpr int kMaxBatches = kUnitTokens * MI_EP_MAX_TOPK / kWave;      // batches of 64 (token, k) pairs in a 64-token unit

template <bool I32>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void layout_hist_kernel(
    const void *__restrict__ topk_idx, int T, int K, int E, int W, int32_t *__restrict__ is_token_in_rank,
    int32_t *__restrict__ unit_hist /*[U][E]*/, int32_t *__restrict__ unit_rank /*[U][W]*/)
{
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int wave = threadIdx.x / kWave, lane = lane_id();
    const int unit = blockIdx.x * kWavesPerBlock + wave;
    int32_t *hist = smem + wave * E;                                       // [E]
    unsigned long long *rmask = (unsigned long long *)(smem + kWavesPerBlock * E) + wave * kUnitTokens;
    const int U = (T + kUnitTokens - 1) / kUnitTokens;
    if (unit >= U) return;   // whole wave exits; no block barriers are used below
    for (int e = lane; e < E; e += kWave) hist[e] = 0;
    rmask[lane] = 0ull;
    const int t0 = unit * kUnitTokens;
    const int ntok = min(kUnitTokens, T - t0);
    const int L = E / W;
    const float inv_k = 1.0f / (float)K, inv_l = 1.0f / (float)L;
    const long long p0 = (long long)t0 * K;
    const int npairs = ntok * K;
    // all of the unit's expert ids first (at most 64 tokens x 16 selections = 16 per lane; index clamped, so unconditional): read inside
    // the loop each batch of 64 was a dependent memory round trip -- eight of them at top-8, most of this kernel's 5.7 us
    long long ev[kMaxBatches];
#pragma unroll
    for (int i = 0; i < kMaxBatches; ++i) ev[i] = load_idx<I32>(topk_idx, p0 + min(i * kWave + lane, npairs - 1));
#pragma unroll
    for (int i = 0; i < kMaxBatches; ++i) {
        const int p = i * kWave + lane;
        if (i * kWave < npairs && p < npairs) {
            const long long e = ev[i];
            if (e >= 0 && e < E) {
                atomicAdd(&hist[(int)e], 1);
                atomicOr(&rmask[div_small(p, inv_k)], 1ull << div_small((int)e, inv_l));
            }
        }
    }
    // single wave: LDS operations above are complete in program order
    for (int e = lane; e < E; e += kWave) unit_hist[(long long)unit * E + e] = hist[e];
    unsigned long long m = (lane < ntok) ? rmask[lane] : 0ull;
    if (lane < ntok) {
        int32_t *row = is_token_in_rank + (long long)(t0 + lane) * W;
        for (int r = 0; r < W; ++r) row[r] = (int32_t)((m >> r) & 1ull);
    }
    for (int r = 0; r < W; ++r) {
        unsigned long long b = __ballot((m >> r) & 1ull);
        if (lane == 0) unit_rank[(long long)unit * W + r] = __popcll(b);
    }
}

__global__ __launch_bounds__(1024) void layout_scan_kernel(int U, int E, int W, const int32_t *__restrict__ unit_hist,
                                                           int32_t *__restrict__ unit_base, const int32_t *__restrict__ unit_rank,
                                                           int32_t *__restrict__ num_tokens_per_rank,
                                                           int32_t *__restrict__ num_tokens_per_expert,
                                                           int32_t *__restrict__ send_data_offset)
{
    __shared__ int32_t wave_tot[16];
    __shared__ int32_t carry;
    __shared__ int32_t part_sum[1024];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid / kWave;
    if (tid == 0) carry = 0;
    __syncthreads();
    if (E <= 512) {
        // E <= 512: `parts` threads share an expert, each owning a contiguous range of units, so the walk over the units is
        // one or two batches of independent loads instead of U / 16 dependent ones (at 4096 tokens, 256 experts: 4 batches -> 1,
        // 8 -> 3 us).  Integer sums: the split changes nothing in the results.
        const int parts = 1024 / E, upp = (U + parts - 1) / parts;      // units per part
        const int e = tid % E, part = tid / E;
        const bool mine = part < parts;
        const int u_lo = part * upp, u_hi = min(U, u_lo + upp);
        int32_t local = 0;
        if (mine)
            for (int u0 = u_lo; u0 < u_hi; u0 += 16) {
                int32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = (u0 + j < u_hi) ? unit_hist[(long long)(u0 + j) * E + e] : 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) local += v[j];
            }
        if (mine) part_sum[part * E + e] = local;
        __syncthreads();
        int32_t run = 0, total = 0;
        if (mine)
            for (int q = 0; q < parts; ++q) {
                const int32_t v = part_sum[q * E + e];
                if (q < part) run += v;
                total += v;
            }
        if (mine)
            for (int u0 = u_lo; u0 < u_hi; u0 += 16) {      // second walk: the loads hit L2, the stores carry the running base
                int32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = (u0 + j < u_hi) ? unit_hist[(long long)(u0 + j) * E + e] : 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (u0 + j < u_hi) unit_base[(long long)(u0 + j) * E + e] = run;
                    run += v[j];
                }
            }
        // exclusive scan of the totals over experts: threads 0 .. E-1 (part 0) hold expert tid
        const bool lead = tid < E;
        const int32_t tot = lead ? total : 0;
        if (lead) num_tokens_per_expert[e] = tot;
        int32_t inc = tot;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            int32_t n = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += n;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        int32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        if (lead) send_data_offset[e] = wbase + inc - tot;
    } else
    for (int base = 0; base < E; base += blockDim.x) {
        const int e = base + tid;
        int32_t run = 0;
        if (e < E) {
            for (int u0 = 0; u0 < U; u0 += 16) {      // 16 independent loads in flight, then the short dependent scan
                int32_t v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = (u0 + j < U) ? unit_hist[(long long)(u0 + j) * E + e] : 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (u0 + j < U) unit_base[(long long)(u0 + j) * E + e] = run;   // the unit's first slot for expert e
                    run += v[j];
                }
            }
            num_tokens_per_expert[e] = run;
        }
        // block exclusive scan of `run` over experts
        int32_t inc = run;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            int32_t n = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += n;
        }
        if (lane == kWave - 1) wave_tot[wave] = inc;
        __syncthreads();
        int32_t wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        const int32_t c = carry;
        if (e < E) send_data_offset[e] = c + wbase + inc - run;
        __syncthreads();
        if (tid == blockDim.x - 1) carry = c + wbase + inc;
        __syncthreads();
    }
    for (int r = tid; r < W; r += blockDim.x) {
        int32_t s = 0;
#pragma unroll 16
        for (int u = 0; u < U; ++u) s += unit_rank[(long long)u * W + r];
        num_tokens_per_rank[r] = s;
    }
}

template <bool I32>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void layout_assign_kernel(
    const void *__restrict__ topk_idx, int T, int K, int E, int nbits, const int32_t *__restrict__ unit_base /*[U][E]*/,
    int32_t *__restrict__ send_token_idx_small)
{
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int wave = threadIdx.x / kWave, lane = lane_id();
    const int unit = blockIdx.x * kWavesPerBlock + wave;
    const int U = (T + kUnitTokens - 1) / kUnitTokens;
    if (unit >= U) return;
    int32_t *cnt = smem + wave * E;
    for (int e = lane; e < E; e += kWave) cnt[e] = unit_base[(long long)unit * E + e];
    const int t0 = unit * kUnitTokens;
    const int npairs = min(kUnitTokens, T - t0) * K;
    const long long p0 = (long long)t0 * K;
    const unsigned long long lt = (1ull << lane) - 1ull;
    long long ev[kMaxBatches];                                  // as in layout_hist_kernel: every id of the unit requested up front
#pragma unroll
    for (int i = 0; i < kMaxBatches; ++i) ev[i] = load_idx<I32>(topk_idx, p0 + min(i * kWave + lane, npairs - 1));
#pragma unroll
    for (int i = 0; i < kMaxBatches; ++i) {
        const int c = i * kWave;
        if (c >= npairs) break;                                 // wave-uniform
        const int p = c + lane;
        const long long e = p < npairs ? ev[i] : -1;
        const bool valid = (e >= 0 && e < E);
        const unsigned long long same = match_any_bits(valid ? (unsigned)e : 0u, valid, nbits);
        int32_t out = 0;
        if (valid) {
            const int before = __popcll(same & lt);
            const int32_t base = cnt[(int)e];
            out = base + before;
            // the highest lane of the group publishes the new running count (LDS ops of one wave are in order,
            // but the read above and this write belong to different lanes of the same instruction pair:
            // every lane reads first because the write below is a later instruction)
            if ((same >> lane) == 1ull) cnt[(int)e] = base + before + 1;
        }
        if (p < npairs) send_token_idx_small[p0 + p] = out;
    }
}

// The three passes in ONE launch: workgroups of 16 waves, a wave per unit (16 units = 1024 tokens per workgroup with 64-token units;
// 16-token units for <= 256 tokens so that a 128-token decode batch occupies 8 waves instead of 2); the per-unit histograms never
// leave LDS.  Same arithmetic and the same deterministic slot order as the three kernels above.
//   one workgroup  (T <= 1024: decode / low-latency mode): removes two launches (~10 us of a ~60 us low-latency dispatch);
//   B workgroups   (larger batches, `sync` != NULL): every workgroup publishes the histogram of ITS 16 units (E + W words), all meet at
//     a grid barrier (B <= 128 co-resident workgroups; two self-resetting words in caller-owned zero-initialised memory), and each
//     derives the running base of its units from the totals of the workgroups in front of it: three launches (4.9 + 6.2 + 6.9 us
//     back to back at 4096 tokens) become one, and the [U][E] histograms / bases never travel through global memory.
__device__ __forceinline__ void layout_grid_barrier(uint32_t *sync, int B)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t0 = wall_clock64();
        while (__hip_atomic_load(sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)B) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > 200000000ull) break;            // 2 s: never hang (a lost launch leaves garbage tables, not a stuck GPU)
        }
        // the last workgroup to LEAVE the spin re-arms both words for the next launch on this stream
        if (__hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)B - 1u) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

template <bool I32, int UT>
__global__ __launch_bounds__(1024) void layout_small_kernel(
    const void *__restrict__ topk_idx, int T, int K, int E, int W, int nbits, int32_t *__restrict__ num_tokens_per_rank,
    int32_t *__restrict__ num_tokens_per_expert, int32_t *__restrict__ is_token_in_rank,
    int32_t *__restrict__ send_token_idx_small, int32_t *__restrict__ send_data_offset,
    int32_t *__restrict__ block_tot /*[B][E + W], B > 1 only*/, uint32_t *__restrict__ sync)
{
    extern __shared__ __attribute__((aligned(16))) int32_t smem[];
    const int B = gridDim.x, blk = blockIdx.x;
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
        layout_grid_barrier(sync, B);
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
        int32_t inc = run;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int32_t n = __shfl_up(inc, off, kWave);
            if (lane >= off) inc += n;
        }
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

}  // namespace mi_ep

using namespace mi_ep;

extern "C" size_t mi_ep_dispatch_layout_workspace(int T, int K, int E)
{
    (void)K;
    size_t U = (size_t)(T + kUnitTokens - 1) / kUnitTokens;
    if (U == 0) U = 1;
    return U * (2 * (size_t)E + MI_EP_MAX_RANKS) * sizeof(int32_t);
}

extern "C" int mi_ep_dispatch_layout(const void *topk_idx, int idx_is_i32, int T, int K, int E, int W,
                                     int32_t *num_tokens_per_rank, int32_t *num_tokens_per_expert,
                                     int32_t *is_token_in_rank, int32_t *send_token_idx_small,
                                     int32_t *send_data_offset, void *workspace, size_t workspace_bytes, uint32_t *sync_words, void *stream)
{
    if (T < 0 || K <= 0 || K > MI_EP_MAX_TOPK || E <= 0 || W <= 0 || W > MI_EP_MAX_RANKS || E % W != 0 || E > 2048)
        return MI_EP_EINVAL;
    if (workspace_bytes < mi_ep_dispatch_layout_workspace(T, K, E) || !workspace) return MI_EP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int U = (T + kUnitTokens - 1) / kUnitTokens;
    int32_t *unit_hist = (int32_t *)workspace;
    int32_t *unit_base = unit_hist + (size_t)(U ? U : 1) * E;
    int32_t *unit_rank = unit_base + (size_t)(U ? U : 1) * E;
    const int blocks = (U + kWavesPerBlock - 1) / kWavesPerBlock;
    const size_t lds1 = (size_t)kWavesPerBlock * E * 4 + (size_t)kWavesPerBlock * kUnitTokens * 8;
    const size_t lds3 = (size_t)kWavesPerBlock * E * 4;
    int nbits = 1;
    while ((1 << nbits) < E) ++nbits;
    // one launch instead of three for decode-size batches; at 4096 tokens the single workgroup (one CU walking 64 units)
    // measured ~30 us slower than the three parallel kernels, so larger batches keep those.  (Folding the scan over the units into
    // the assign kernel -- every wave summing the histograms in front of its own unit -- was also built: bit-exact, two launches
    // instead of three, and 5.7 + 12.2 us instead of 5.7 + 5.9 + 7.4 us back to back: not worth its code.)
    // one launch: a single workgroup for <= 16 units; for more, the cooperative form when the caller lends two persistent sync words
    // (MI_EP_LAYOUT_COOP=0 keeps the three launches) and the grid is small enough to be co-resident whatever else runs.  Unit size:
    // 16 tokens wherever the grid allows (<= 32768 tokens): a unit's serial chain -- LDS count read -> ballots -> count write, once per
    // batch of 64 pairs -- is 2 batches long instead of 8 at top-8 (pass 1 + pass 3 of a 64-token unit: 6.5 + 4.4 us, of a 16-token
    // unit 2.5 + 1.2 us, tools/probes/time_layout_phases.py); 64-token units beyond that.
    static const bool coop_ok = !(getenv("MI_EP_LAYOUT_COOP") && atoi(getenv("MI_EP_LAYOUT_COOP")) == 0);
    const bool lend = sync_words && coop_ok;
    int ut = kUnitTokens;
    if (T <= 256 || (lend && T <= 128 * 256)) ut = 16;
    const int Us = (T + ut - 1) / ut;
    const int Bc = (Us + 15) / 16;                         // workgroups of 16 units
    const bool single = Us >= 1 && Us <= 16;
    const bool coop = Us > 16 && lend && Bc <= 128 && workspace_bytes >= (size_t)Bc * (E + MI_EP_MAX_RANKS) * sizeof(int32_t);
    if ((single || coop) && (size_t)16 * E <= 16384 && ((16 * E) & 1) == 0) {
        const size_t ldsf = (size_t)16 * E * 4 + (size_t)16 * ut * 8 + (size_t)(W + 16 + 4) * 4;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
#define MI_EP_LAYOUT_SMALL(I32, UT)                                                                                              \
    layout_small_kernel<I32, UT><<<Bc, 1024, ldsf, s>>>(topk_idx, T, K, E, W, nbits, num_tokens_per_rank, num_tokens_per_expert, \
                                                        is_token_in_rank, send_token_idx_small, send_data_offset,                \
                                                        (int32_t *)workspace, sync_words)
        if (ut == 16) { if (idx_is_i32) MI_EP_LAYOUT_SMALL(true, 16); else MI_EP_LAYOUT_SMALL(false, 16); }
        else { if (idx_is_i32) MI_EP_LAYOUT_SMALL(true, 64); else MI_EP_LAYOUT_SMALL(false, 64); }
#undef MI_EP_LAYOUT_SMALL
        return launch_status();
    }
    if (U > 0) {
        if (idx_is_i32)
            layout_hist_kernel<true><<<blocks, kWave * kWavesPerBlock, lds1, s>>>(topk_idx, T, K, E, W, is_token_in_rank,
                                                                                  unit_hist, unit_rank);
        else
            layout_hist_kernel<false><<<blocks, kWave * kWavesPerBlock, lds1, s>>>(topk_idx, T, K, E, W, is_token_in_rank,
                                                                                   unit_hist, unit_rank);
    }
    layout_scan_kernel<<<1, 1024, 0, s>>>(U, E, W, unit_hist, unit_base, unit_rank, num_tokens_per_rank, num_tokens_per_expert,
                                          send_data_offset);
    if (U > 0) {
        if (idx_is_i32)
            layout_assign_kernel<true><<<blocks, kWave * kWavesPerBlock, lds3, s>>>(topk_idx, T, K, E, nbits, unit_base,
                                                                                    send_token_idx_small);
        else
            layout_assign_kernel<false><<<blocks, kWave * kWavesPerBlock, lds3, s>>>(topk_idx, T, K, E, nbits, unit_base,
                                                                                     send_token_idx_small);
    }
    return launch_status();
}
