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
#include "device_once.h"
#include "ep_common.h"
#include "layout_dev.h"

namespace mi_ep {

constexpr int kUnitTokens = 64;
constexpr int kWavesPerBlock = 4;

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
                                     int32_t *send_data_offset, void *workspace, size_t workspace_bytes, uint32_t *sync_words, int32_t *status,
                                     void *stream)
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
        const size_t ldsf = layout_small_lds_bytes(E, W, ut);
        static PerDeviceOnce attr_once;
        if (attr_once.need()) {
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void *)layout_small_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
#define MI_EP_LAYOUT_SMALL(I32, UT)                                                                                              \
    layout_small_kernel<I32, UT><<<Bc, 1024, ldsf, s>>>(topk_idx, T, K, E, W, nbits, num_tokens_per_rank, num_tokens_per_expert, \
                                                        is_token_in_rank, send_token_idx_small, send_data_offset,                \
                                                        (int32_t *)workspace, sync_words, status)
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

// ---- shared-expert ranks (MOE_SHARED_EXPERT_RANK_NUM = S > 0) ------------------------------------------------------------------------
// Reference: the first S ranks of the group hold ONE expert each (the shared expert), the other W - S ranks hold L = E / (W - S) routed
// experts each (deep_ep.cpp:866-874, 1219-1220); every token that has at least one active selection is ALSO sent to shared rank
// (my_rank mod S), at its position among those tokens, with k = K in its triple (moe_distribute_dispatch_v2.h:555-604, 748-779, 918-960),
// and the combine adds that row unweighted after the K weighted ones (moe_distribute_combine_v2.h:1219-1235).  Here that is a renaming of
// experts in front of the ordinary kernels: W ranks x L expert slots, routed expert e -> slot S*L + e (rank S + e / L, local e mod L), the
// shared expert of rank s -> slot s*L (shared ranks use local slot 0 only), as a (K+1)-th selection of weight 1 (x * 1.0f is exact).
template <bool I32>
__global__ __launch_bounds__(256) void shared_expert_map_kernel(const void *__restrict__ topk_idx, const float *__restrict__ w_in, int T, int K, int E, int S,
                                                                int L, int my_rank, int32_t *__restrict__ idx_out, float *__restrict__ w_out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    bool any = false;
    for (int k = 0; k < K; ++k) {
        const long long e = I32 ? (long long)((const int32_t *)topk_idx)[(size_t)t * K + k] : ((const long long *)topk_idx)[(size_t)t * K + k];
        const bool ok = e >= 0 && e < E;
        any |= ok;
        idx_out[(size_t)t * (K + 1) + k] = ok ? (int32_t)(e + (long long)S * L) : -1;
        if (w_out) w_out[(size_t)t * (K + 1) + k] = w_in ? w_in[(size_t)t * K + k] : 1.0f;
    }
    idx_out[(size_t)t * (K + 1) + K] = any ? (my_rank % S) * L : -1;
    if (w_out) w_out[(size_t)t * (K + 1) + K] = 1.0f;
}

extern "C" int mi_ep_shared_expert_map(const void *topk_idx, int idx_is_i32, const float *topk_weights, int T, int K, int E, int W, int S,
                                       int my_rank, int32_t *idx_out, float *weights_out, void *stream)
{
    if (T < 0 || K <= 0 || K >= MI_EP_MAX_TOPK || E <= 0 || S <= 0 || S >= W || E % (W - S) || my_rank < 0 || my_rank >= W) return MI_EP_EINVAL;
    if (T == 0) return MI_EP_OK;
    if (!topk_idx || !idx_out) return MI_EP_EINVAL;
    const int L = E / (W - S);
    hipStream_t s = (hipStream_t)stream;
    if (idx_is_i32) shared_expert_map_kernel<true><<<(T + 255) / 256, 256, 0, s>>>(topk_idx, topk_weights, T, K, E, S, L, my_rank, idx_out, weights_out);
    else shared_expert_map_kernel<false><<<(T + 255) / 256, 256, 0, s>>>(topk_idx, topk_weights, T, K, E, S, L, my_rank, idx_out, weights_out);
    return launch_status();
}
