// A4 / A6 combine for gfx950: push (expert side writes each BF16 row into the source rank's slot
// t*K+k) and reduce (owner: fp32 weighted sum of the K slots of a token, k ascending, -> bf16).
// Replaces aclnnCamMoeCombineNormal / aclnnMoeLowLatencyCombineV2 (reference kernels
// csrc/deepep/ops/op_kernel/cam_moe_combine_normal.h:291-321 (push), :359-400 (weighted sum);
// csrc/deepep/ops/op_kernel/moe_distribute_combine_v2.h:818-851,1102-1256).
//
// MI355X design: one wave64 per row for the push (16-B/lane stores, 8 in flight per lane; rows are in
// (local expert, src rank) order so consecutive workgroups target different peers / xGMI links);
// the reduce gives each wave a (token, 512-element segment) tile: K x 16-B loads in flight per lane,
// separate v_mul_f32 / v_add_f32 (file is built with -ffp-contract=off) to reproduce the reference's
// Muls + Add ordering bit for bit.  Per-rank HBM traffic: push R*2H read + R*2H write, reduce
// T*K*2H read + T*2H write.
#include <stdlib.h>

#include <algorithm>

#include "ep_common.h"
#include <type_traits>

namespace mi_ep {

constexpr int kPushWaves = 4;

// "My rows are pushed" raised from INSIDE the push launch (TAIL): every workgroup writes its rows through (sc0 sc1), drains, and counts
// itself in at a device word of the rank's control area; the last one to arrive does what signal_wait_kernel does in a launch of its
// own -- raise this rank's flag at every owner, wait (bounded) for every expert rank's, complete the family's call counter -- and
// re-arms the word.  One launch and one kernel boundary less per combine (low-latency: three launches -> two).
struct PushTail {
    uint32_t *arrive;             // zero between calls (control area: zeroed at creation, re-armed by the last arriver)
    PeerPtrs flag_peers;          // every rank's combine flag group
    const uint64_t *my_flags;
    uint64_t *epoch_bump;         // the family's completed-call counter (also this call's epoch source: counter + 1)
    int32_t *status;
    uint64_t timeout_ticks;
    // FLAGGED form (arrive == NULL, row_flag_peers set): no tail at all -- every wave raises the flag word of ITS row at the row's owner once
    // the row's write-through stores have drained (the reference's per-token arrival state, moe_distribute_combine_v2.h:952-1002); the
    // owner's reduce waits per selection (combine_reduce_kernel<.., FLAGGED>).  Words are tagged with the call's epoch, never cleared.
    PeerPtrs row_flag_peers;      // every rank's row-flag area: uint32 [2 halves][slot rows]
    size_t row_flags_parity_stride;
    uint64_t *cur_epoch;          // FLAGGED: the push leaves this call's epoch here for the reduce (which then completes the call counter)
};

template <bool TAIL>
__global__ __launch_bounds__(kWave * kPushWaves) void combine_push_kernel(
    const uint8_t *__restrict__ x, const int32_t *__restrict__ src_idx, const int32_t *__restrict__ total_dev,
    int rows_hint, int row_bytes /*2H*/, size_t slot_stride, int K, int W, PeerPtrs dsts, Parity par, int slot_rows, int my_rank,
    int32_t *__restrict__ local_row, unsigned deal_stride, PushTail tail)
{
    // never trust the device-side count beyond the rows the caller's tensor holds
    const int total = total_dev ? min(*total_dev, rows_hint) : rows_hint;
    if (TAIL && tail.cur_epoch && blockIdx.x == 0 && threadIdx.x == 0) *tail.cur_epoch = *tail.epoch_bump + 1ull;
    const size_t poff = parity_off(par);
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const int n16 = row_bytes / 16;
    {   // rows dealt to waves round-robin over the whole grid (see pull_body).  Rows arrive in (local expert, source rank) order, so waves
        // i, i + 1, ... of the plain deal write to the SAME peer for a whole segment (~R / (L W) rows: 128 at C2) -- four waves of a CU,
        // and at small batches the whole chip, on one xGMI link at a time.  With W > 1 the deal is therefore scattered: wave step i takes
        // row (i * deal_stride) mod 2^k (odd stride: a bijection on [0, 2^k), 2^k >= total; the indices >= total are skipped), so that
        // neighbouring waves sit two segments apart.  Every row is still copied exactly once into its own slot: bit-neutral.
        unsigned pow2m1 = 0;
        if (deal_stride > 1) {
            pow2m1 = (unsigned)max(total, 1) - 1u;
            pow2m1 |= pow2m1 >> 1, pow2m1 |= pow2m1 >> 2, pow2m1 |= pow2m1 >> 4, pow2m1 |= pow2m1 >> 8, pow2m1 |= pow2m1 >> 16;
        }
        const long long limit = deal_stride > 1 ? (long long)pow2m1 + 1 : (long long)total;
#pragma unroll 1
        for (long long i = (long long)blockIdx.x * kPushWaves + wave; i < limit; i += (long long)gridDim.x * kPushWaves) {
            const long long r = deal_stride > 1 ? (long long)(((unsigned)i * deal_stride) & pow2m1) : i;
            if (r >= total) continue;
            const int src = src_idx[r * 3 + 0];
            const int t = src_idx[r * 3 + 1];
            const int k = src_idx[r * 3 + 2];
            // corrupted / mismatched handle: drop the row instead of a wild (cross-GPU) store
            if (src < 0 || src >= W || k < 0 || k >= K || t < 0 || (long long)t * K + k >= slot_rows) continue;
            // a row whose token lives on this rank does not travel: the reduce reads it where it is (x), all it needs is the row number
            if (local_row && src == my_rank) {
                if (lane == 0) local_row[(long long)t * K + k] = (int32_t)r;
                continue;
            }
            const u32x4 *s16 = (const u32x4 *)(x + (size_t)r * row_bytes);
            u32x4 *d16 = (u32x4 *)((uint8_t *)dsts.p[src] + poff + ((size_t)t * K + k) * slot_stride);
            // straight-line groups of whole 1 KB pieces (copy_row, ep_common.h): loads back to back, then the (possibly remote) stores.
            // (nontemporal stores were measured: the push slows 159 -> 183 us, the reduce that follows speeds up 137 -> 122 us because
            //  fewer dirty lines are left behind -- a wash for the step)
            if (TAIL) copy_row_wt<true>(s16, d16, n16, lane);
            else copy_row<true, false>(s16, d16, n16, lane);
            if (TAIL && !tail.arrive) {
                // FLAGGED: the row is at its owner (write-through stores, drained) before its flag says so
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const uint64_t ep64 = *tail.epoch_bump + 1ull;
                    uint32_t *fl = (uint32_t *)((uint8_t *)tail.row_flag_peers.p[src] + (size_t)(ep64 & 1ull) * tail.row_flags_parity_stride) +
                                   ((size_t)t * K + k);
                    __hip_atomic_store(fl, (uint32_t)ep64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    if (TAIL && tail.arrive) {
        __shared__ uint32_t last_s;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave: its write-through stores are performed
        __syncthreads();
        if (threadIdx.x == 0) {
            // (the local_row words, plain stores read by the NEXT launch, need nothing: the kernel boundary publishes them)
            const uint32_t old = __hip_atomic_fetch_add(tail.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = old == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (last_s == 0) return;
        // the last workgroup: everybody's rows are out.  One wave: signal + wait (signal_wait_kernel, sync.hip), then re-arm.
        if (threadIdx.x < kWave) {
            const int s = threadIdx.x;
            const uint64_t epoch = *tail.epoch_bump + 1ull;
            if (s < W) {
                sys_store_u64((uint64_t *)tail.flag_peers.p[s] + my_rank, epoch);
                const uint64_t t0 = ticks_100mhz();
                while (sys_load_u64(tail.my_flags + s) < epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (ticks_100mhz() - t0 > tail.timeout_ticks) {
                        report_status(tail.status, 1 + s);
                        break;
                    }
                }
            }
            // one wave: every lane has read the counter before lane 0 moves it (wave-level program order)
            if (s == 0) {
                *tail.epoch_bump = epoch;
                __hip_atomic_store(tail.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// FLAGGED (the two-launch combine): the rows were pushed by mi_ep_combine_push_flagged; a wave waits (bounded) for the flag words of ITS token's
// selections before it reads their rows, and the last workgroup of the launch to finish completes the family's call counter.
struct ReduceFlags {
    const uint32_t *row_flags;    // this rank's row-flag area (both halves)
    size_t parity_stride;
    const uint64_t *cur_epoch;    // this call's epoch, left by the push launch (nothing in this launch reads the completed-call counter ...
    uint64_t *epoch_bump;         // ... so its first workgroup completes it without waiting for anybody)
    int32_t *status;
    uint64_t timeout_ticks;
    int max_blocks;               // host side only: > 0 caps the launch (ranks sharing one GPU)
};
template <bool I32, int KMAX, bool FLAGGED>
__device__ __forceinline__ void combine_reduce_body(
    const uint8_t *__restrict__ slots, size_t slot_stride, const void *__restrict__ topk_idx,
    const float *__restrict__ topk_w, const int32_t *__restrict__ send_off, const int32_t *__restrict__ idx_small,
    int T, int K, int H, int E, int segs_per_token, uint16_t *__restrict__ out, const Parity &par, const uint8_t *__restrict__ x_local,
    const int32_t *__restrict__ local_row, int local_rows, int my_rank, int experts_per_rank, const ReduceFlags &rf)
{
    slots += parity_off(par);
    const int lane = lane_id();
    // wave-uniform 32-bit index arithmetic (the launcher bounds T * segs_per_token): a 64-bit division per lane was ~150 VALU operations
    const uint32_t wid = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave);
    const uint32_t t = wid / (uint32_t)segs_per_token;
    const int seg0 = (int)(wid - t * (uint32_t)segs_per_token);
    if (t >= (uint32_t)T) return;
    // routing + weights of this token (lane k < K); lane k also forms the address of ITS row, so that the K row bases reach the loads
    // below as wave-uniform values (v_readlane -> SGPR pair + one lane offset) instead of K 64-bit multiplications and selects per lane
    float w_l = 0.f;
    bool valid_l = false;
    // absent / invalid selections re-read a row that is always there and is never summed: slot t*K of the window layout (T*K slots), row 0
    // of the all-to-all return buffer (send_off != NULL: it holds exactly the valid pairs, at least one row -- row t*K may lie past its end)
    const uint8_t *base_l = send_off ? slots : slots + (size_t)t * K * slot_stride;
    if (lane < K) {
        const size_t tk = (size_t)t * K + lane;
        long long e = I32 ? (long long)((const int32_t *)topk_idx)[tk] : ((const long long *)topk_idx)[tk];
        valid_l = (e >= 0 && e < E);
        w_l = topk_w ? topk_w[tk] : 1.0f;
        long long slot_l = (long long)tk;      // slot mode: t*K+k (window push) or the dispatch send slot (all-to-all return)
        if (send_off && valid_l) slot_l = (long long)send_off[e] + idx_small[tk];
        if (valid_l) base_l = slots + (size_t)(int)slot_l * slot_stride;
        // selections served by this rank's own experts were not pushed: their rows are read from the expert output itself
        bool local_l = false;
        if (x_local && valid_l && (int)((uint32_t)e / (uint32_t)experts_per_rank) == my_rank) {    // 0 <= e < E: 32-bit division
            base_l = x_local + (size_t)min(max(local_row[tk], 0), local_rows - 1) * ((size_t)H * 2);       // never read outside x, whatever the handle says
            local_l = true;
        }
        if constexpr (FLAGGED) {
            // the row of selection (t, lane) has landed when its flag word carries this call's epoch (bounded wait: a missing row is reported
            // and the slot is summed as it is)
            if (valid_l && !local_l) {
                const uint64_t ep64 = *rf.cur_epoch;
                const uint32_t *fl = (const uint32_t *)((const uint8_t *)rf.row_flags + (size_t)(ep64 & 1ull) * rf.parity_stride) + tk;
                const uint64_t t0 = ticks_100mhz();
                while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)ep64) {
                    __builtin_amdgcn_s_sleep(2);
                    if (ticks_100mhz() - t0 > rf.timeout_ticks) {
                        report_status(rf.status, 3000 + lane);
                        break;
                    }
                }
            }
        }
    }
    // (FLAGGED: the row loads below are issued after the flag loads above have returned -- the wait loop's exit depends on them -- and are
    //  SYSTEM-scope loads like the poll (ld_sys_b128): slot rows are 16-byte aligned, so a neighbour's read, or this half's use two calls ago,
    //  may have left a line of this row in a cache, which a plain / nontemporal load could be answered from.  An acquire fence here instead
    //  -- a cache invalidate per wave -- doubled the launch.)
    if constexpr (FLAGGED) asm volatile("" ::: "memory");
    const unsigned long long vmask = __ballot(valid_l);
    const uint64_t base_bits = (uint64_t)(uintptr_t)base_l;
    const int base_lo = (int)(uint32_t)base_bits, base_hi = (int)(uint32_t)(base_bits >> 32);
    float w[KMAX];
    const uint8_t *rowp[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        w[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_l), k));
        rowp[k] = (const uint8_t *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(base_hi, k) << 32) |
                                               (uint32_t)__builtin_amdgcn_readlane(base_lo, k));
    }
    const int nchunks = H / 8;          // 16-B chunks of 8 bf16
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) u32x4 *gptr;       // global, not generic: the bases went through v_readlane as integers
    // Two elements per VALU instruction (v_pk_mul_f32, v_pk_add_f32: separately rounded, as the scalar pair was) and the hardware bf16
    // rounding: a wave instruction takes 4 cycles on the 16-lane SIMDs, and the scalar form kept them ~40 % busy (104 -> 93 us at C2).
    // read-once stream: nontemporal loads keep the 0.47 GB of slots out of L2/MALL (measured 141 -> 102 us at C2)
    __amdgpu_buffer_rsrc_t rs[FLAGGED ? KMAX : 1];
    if constexpr (FLAGGED) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) rs[k] = sys_row_rsrc(rowp[k], H * 2);
    }
    auto chunk = [&](int c, auto all_tag) {
        constexpr bool ALL = decltype(all_tag)::value;
        u32x4 v[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            // unconditional in both paths: an absent / invalid selection re-reads a row that is always there (base_l above) and is skipped in
            // the sum -- under the wave-uniform condition each load got its own block and its own vmcnt(0)
#if defined(MI_COMBINE_FLAGGED_NT)          // timing probe only: the loads the unflagged form uses
            if constexpr (FLAGGED) v[k] = __builtin_nontemporal_load((gptr)(uintptr_t)(rowp[k] + (uint32_t)c * 16u));
#elif defined(MI_COMBINE_FLAGGED_SC1)       // timing probe only: device-scope instead of system-scope loads
            if constexpr (FLAGGED) v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], (int)((uint32_t)c * 16u), 0, 16));
#else
            if constexpr (FLAGGED) v[k] = ld_sys_b128(rs[k], (uint32_t)c * 16u);
#endif
            else v[k] = __builtin_nontemporal_load((gptr)(uintptr_t)(rowp[k] + (uint32_t)c * 16u));
        }
        f32x2 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x2{0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (ALL || (k < K && ((vmask >> k) & 1ull))) {
                const f32x2 wk = f32x2{w[k], w[k]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 val = f32x2{bf16_to_f32(v[k][j] & 0xFFFFu), __uint_as_float(v[k][j] & 0xFFFF0000u)};
                    acc[j] = acc[j] + val * wk;
                }
            }
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(acc[j], bf16x2));
        *(u32x4 *)(out + (size_t)t * H + (size_t)c * 8) = o;
    };
    // Every selection of the token present (the usual case): straight-line code.  With -1 entries or K < KMAX (top-6 models on the
    // top-8 build) the sum skips the absent rows under wave-uniform branches; the KMAX row reads are issued back to back either way.
    if (K == KMAX && vmask == (KMAX == 64 ? ~0ull : (1ull << KMAX) - 1ull)) {
        for (int c = seg0 * kWave + lane; c < nchunks; c += segs_per_token * kWave) chunk(c, std::true_type{});
    } else {
        for (int c = seg0 * kWave + lane; c < nchunks; c += segs_per_token * kWave) chunk(c, std::false_type{});
    }
}

template <bool I32, int KMAX, bool FLAGGED = false>
__global__ __launch_bounds__(256) void combine_reduce_kernel(
    const uint8_t *__restrict__ slots, size_t slot_stride, const void *__restrict__ topk_idx,
    const float *__restrict__ topk_w, const int32_t *__restrict__ send_off, const int32_t *__restrict__ idx_small,
    int T, int K, int H, int E, int segs_per_token, uint16_t *__restrict__ out, Parity par, const uint8_t *__restrict__ x_local,
    const int32_t *__restrict__ local_row, int local_rows, int my_rank, int experts_per_rank, ReduceFlags rf)
{
    combine_reduce_body<I32, KMAX, FLAGGED>(slots, slot_stride, topk_idx, topk_w, send_off, idx_small, T, K, H, E, segs_per_token, out, par, x_local,
                                            local_row, local_rows, my_rank, experts_per_rank, rf);
    if constexpr (FLAGGED) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *rf.epoch_bump = *rf.cur_epoch;
    }
}

// Expert side of the all-to-all transport: reorder rows from dispatch order (local expert, src rank, j) into
// per-source contiguous blocks (src rank, local expert, j) -- the order in which that source staged them, so the
// block can travel back as one message and lands slot-aligned in the source's return buffer.
__global__ __launch_bounds__(256) void combine_pack_kernel(const uint8_t *__restrict__ x, const int32_t *__restrict__ send_head,
                                                           int W, int L, int row_bytes, uint8_t *__restrict__ packed,
                                                           int32_t *__restrict__ rows_per_src)
{
    extern __shared__ __attribute__((aligned(16))) int32_t sm[];
    const int LW = L * W;
    int32_t *cum = sm;            // [LW] inclusive cumsum (dispatch order)
    int32_t *dstoff = sm + LW;    // [LW] first packed row of segment i
    int32_t *blk = dstoff + LW;   // [W+1] block start per src
    for (int i = threadIdx.x; i < LW; i += blockDim.x) cum[i] = send_head[i];
    __syncthreads();
    if (threadIdx.x < W) {
        const int src = threadIdx.x;
        int32_t s = 0;
        for (int le = 0; le < L; ++le) {
            const int i = le * W + src;
            s += cum[i] - (i ? cum[i - 1] : 0);
        }
        blk[src + 1] = s;
        if (blockIdx.x == 0 && rows_per_src) rows_per_src[src] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        blk[0] = 0;
        for (int src = 0; src < W; ++src) blk[src + 1] += blk[src];
    }
    __syncthreads();
    if (threadIdx.x < W) {
        const int src = threadIdx.x;
        int32_t run = blk[src];
        for (int le = 0; le < L; ++le) {
            const int i = le * W + src;
            dstoff[i] = run;
            run += cum[i] - (i ? cum[i - 1] : 0);
        }
    }
    __syncthreads();
    const int total = cum[LW - 1];
    const int lane = lane_id(), wave = threadIdx.x / kWave, nw = blockDim.x / kWave;
    const int n16 = row_bytes / 16;
    for (long long r = (long long)blockIdx.x * nw + wave; r < total; r += (long long)gridDim.x * nw) {
        int lo = 0, hi = LW - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] > r) hi = mid; else lo = mid + 1;
        }
        const int j = (int)(r - (lo ? cum[lo - 1] : 0));
        const u32x4 *s16 = (const u32x4 *)(x + (size_t)r * row_bytes);
        u32x4 *d16 = (u32x4 *)(packed + ((size_t)dstoff[lo] + j) * row_bytes);
        for (int base = 0; base < n16; base += kWave * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int item = base + u * kWave + lane;
                if (item < n16) v[u] = __builtin_nontemporal_load(s16 + item);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int item = base + u * kWave + lane;
                if (item < n16) d16[item] = v[u];
            }
        }
    }
}

}  // namespace mi_ep

using namespace mi_ep;

extern "C" int mi_ep_combine_pack(const void *x, const int32_t *send_head, int W, int L, int H, int rows_hint,
                                  void *packed, int32_t *rows_per_src, void *stream)
{
    if (!send_head || W <= 0 || W > MI_EP_MAX_RANKS || L <= 0 || L * W > 2048 || H <= 0 || H % 8) return MI_EP_EINVAL;
    if (rows_hint > 0 && (!x || !packed)) return MI_EP_EINVAL;
    long long blocks = rows_hint > 0 ? ((long long)rows_hint + 3) / 4 : 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const size_t lds = (size_t)(2 * L * W + W + 1) * sizeof(int32_t);
    combine_pack_kernel<<<(int)blocks, 256, lds, (hipStream_t)stream>>>((const uint8_t *)x, send_head, W, L, H * 2,
                                                                      (uint8_t *)packed, rows_per_src);
    return launch_status();
}

extern "C" size_t mi_ep_combine_row_bytes(int hidden) { return ((size_t)hidden * 2 + 15) / 16 * 16; }

static int combine_push_launch(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int H, int K,
                               void *const *dst_base_host, int W, size_t slot_region_bytes, const uint64_t *epoch_ctr, size_t parity_stride,
                               int my_rank, int32_t *local_row, const PushTail *tail, void *stream)
{
    if (H <= 0 || H % 8 || K <= 0 || K > MI_EP_MAX_TOPK || W <= 0 || W > MI_EP_MAX_RANKS || !dst_base_host ||
        (local_row && (my_rank < 0 || my_rank >= W)))
        return MI_EP_EINVAL;
    if (rows_hint <= 0 && !tail) return MI_EP_OK;
    if (rows_hint > 0 && (!x || !src_idx)) return MI_EP_EINVAL;
    PeerPtrs pp;
    for (int i = 0; i < W; ++i) {
        if (!dst_base_host[i]) return MI_EP_EINVAL;
        pp.p[i] = dst_base_host[i];
    }
    long long blocks = ((long long)std::max(rows_hint, 1) + kPushWaves - 1) / kPushWaves;        // one row per wave until the chip is full
    static const long long cap = getenv("MI_EP_PUSH_BLOCKS") ? atoll(getenv("MI_EP_PUSH_BLOCKS")) : 256 * 8;
    if (blocks > cap) blocks = cap;
    // MI_EP_PUSH_STRIDE: 1 = rows in order (the W = 1 form), odd > 1 = scattered deal (default 257 at W > 1, see the kernel)
    static const long long stride_env = getenv("MI_EP_PUSH_STRIDE") ? atoll(getenv("MI_EP_PUSH_STRIDE")) : 0;
    unsigned deal_stride = stride_env > 0 ? (unsigned)stride_env | 1u : (W > 1 ? 257u : 1u);
    const Parity par = make_parity(epoch_ctr, 1, parity_stride);
    const int slot_rows = slot_region_bytes ? (int)std::min<size_t>(slot_region_bytes / mi_ep_combine_row_bytes(H), 0x7FFFFFFF) : 0x7FFFFFFF;
    if (tail)
        combine_push_kernel<true><<<(int)blocks, kWave * kPushWaves, 0, (hipStream_t)stream>>>(
            (const uint8_t *)x, src_idx, total_rows_dev, std::max(rows_hint, 0), H * 2, mi_ep_combine_row_bytes(H), K, W, pp, par, slot_rows, my_rank,
            local_row, deal_stride, *tail);
    else
        combine_push_kernel<false><<<(int)blocks, kWave * kPushWaves, 0, (hipStream_t)stream>>>(
            (const uint8_t *)x, src_idx, total_rows_dev, rows_hint, H * 2, mi_ep_combine_row_bytes(H), K, W, pp, par, slot_rows, my_rank, local_row,
            deal_stride, PushTail{});
    return launch_status();
}

extern "C" int mi_ep_combine_push(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint,
                                  int H, int K, void *const *dst_base_host, int W, size_t slot_region_bytes,
                                  const uint64_t *epoch_ctr, size_t parity_stride, int my_rank, int32_t *local_row, void *stream)
{
    return combine_push_launch(x, src_idx, total_rows_dev, rows_hint, H, K, dst_base_host, W, slot_region_bytes, epoch_ctr, parity_stride, my_rank,
                               local_row, nullptr, stream);
}

extern "C" int mi_ep_combine_push_signal_wait(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int H, int K,
                                              void *const *dst_base_host, int W, size_t slot_region_bytes, uint64_t *epoch_ctr,
                                              size_t parity_stride, int my_rank, int32_t *local_row, uint64_t *const *peer_flags_host,
                                              const uint64_t *my_flags, uint32_t *arrive_word, int32_t *status, int timeout_ms, void *stream)
{
    if (!epoch_ctr || !peer_flags_host || !my_flags || !arrive_word || !status || W <= 0 || W > MI_EP_MAX_RANKS || my_rank < 0 || my_rank >= W)
        return MI_EP_EINVAL;
    PushTail tail{};
    tail.arrive = arrive_word, tail.my_flags = my_flags, tail.epoch_bump = epoch_ctr, tail.status = status;
    tail.timeout_ticks = (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull;
    for (int i = 0; i < W; ++i) {
        if (!peer_flags_host[i]) return MI_EP_EINVAL;
        tail.flag_peers.p[i] = peer_flags_host[i];
    }
    return combine_push_launch(x, src_idx, total_rows_dev, rows_hint, H, K, dst_base_host, W, slot_region_bytes, epoch_ctr, parity_stride, my_rank,
                               local_row, &tail, stream);
}

// the push of the two-launch combine: mi_ep_combine_push whose waves raise the flag word of every row they have written at the row's owner
extern "C" int mi_ep_combine_push_flagged(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int H, int K,
                                          void *const *dst_base_host, int W, size_t slot_region_bytes, const uint64_t *epoch_ctr,
                                          size_t parity_stride, int my_rank, int32_t *local_row, uint32_t *const *peer_row_flags_host,
                                          size_t row_flags_parity_stride, uint64_t *cur_epoch_word, void *stream)
{
    if (!epoch_ctr || !peer_row_flags_host || !cur_epoch_word || W <= 0 || W > MI_EP_MAX_RANKS || my_rank < 0 || my_rank >= W || H <= 0 || H % 8 ||
        row_flags_parity_stride < sizeof(uint32_t))
        return MI_EP_EINVAL;
    // (a call without rows still launches one workgroup: it leaves the call's epoch for the reduce)
    // a slot without a flag word is no slot: (t, k) past the words of a half are dropped like those past the region
    const size_t flagged_bytes = (row_flags_parity_stride / sizeof(uint32_t)) * mi_ep_combine_row_bytes(H);
    slot_region_bytes = slot_region_bytes ? std::min(slot_region_bytes, flagged_bytes) : flagged_bytes;
    PushTail tail{};
    tail.epoch_bump = const_cast<uint64_t *>(epoch_ctr), tail.row_flags_parity_stride = row_flags_parity_stride, tail.cur_epoch = cur_epoch_word;
    for (int i = 0; i < W; ++i) {
        if (!peer_row_flags_host[i]) return MI_EP_EINVAL;
        tail.row_flag_peers.p[i] = peer_row_flags_host[i];
    }
    return combine_push_launch(x, src_idx, total_rows_dev, rows_hint, H, K, dst_base_host, W, slot_region_bytes, epoch_ctr, parity_stride, my_rank,
                               local_row, &tail, stream);
}

static int combine_reduce_launch(const void *slots, const void *topk_idx, int idx_is_i32, const float *topk_weights,
                                 const int32_t *send_data_offset, const int32_t *send_token_idx_small, int T, int K,
                                 int H, int E, void *out, const uint64_t *epoch_ctr, size_t parity_stride, const void *x_local,
                                 const int32_t *local_row, int local_rows, int my_rank, int num_ranks, const ReduceFlags *rf, void *stream);

extern "C" int mi_ep_combine_reduce(const void *slots, const void *topk_idx, int idx_is_i32, const float *topk_weights,
                                    const int32_t *send_data_offset, const int32_t *send_token_idx_small, int T, int K,
                                    int H, int E, void *out, const uint64_t *epoch_ctr, size_t parity_stride, const void *x_local,
                                    const int32_t *local_row, int local_rows, int my_rank, int num_ranks, void *stream)
{
    return combine_reduce_launch(slots, topk_idx, idx_is_i32, topk_weights, send_data_offset, send_token_idx_small, T, K, H, E, out, epoch_ctr,
                                 parity_stride, x_local, local_row, local_rows, my_rank, num_ranks, nullptr, stream);
}

// the reduce of the two-launch combine: waits per selection for the row flags mi_ep_combine_push_flagged raises, completes the call counter
extern "C" int mi_ep_combine_reduce_flagged(const void *slots, const void *topk_idx, int idx_is_i32, const float *topk_weights, int T, int K, int H,
                                            int E, void *out, uint64_t *epoch_ctr, size_t parity_stride, const void *x_local,
                                            const int32_t *local_row, int local_rows, int my_rank, int num_ranks, const uint32_t *my_row_flags,
                                            size_t row_flags_parity_stride, const uint64_t *cur_epoch_word, int32_t *status, int timeout_ms,
                                            int max_blocks, void *stream)
{
    if (!epoch_ctr || !my_row_flags || !cur_epoch_word || !status || T < 0 || K <= 0 ||
        (unsigned long long)T * (unsigned long long)K > row_flags_parity_stride / sizeof(uint32_t))
        return MI_EP_EINVAL;
    ReduceFlags rf{};
    rf.row_flags = my_row_flags, rf.parity_stride = row_flags_parity_stride, rf.epoch_bump = epoch_ctr, rf.cur_epoch = cur_epoch_word, rf.status = status;
    rf.timeout_ticks = (uint64_t)(timeout_ms > 0 ? timeout_ms : 10000) * 100000ull;
    rf.max_blocks = max_blocks;
    // (the ping-pong half comes from the epoch word the push left, not from the completed-call counter this launch moves)
    return combine_reduce_launch(slots, topk_idx, idx_is_i32, topk_weights, nullptr, nullptr, T, K, H, E, out, cur_epoch_word, parity_stride, x_local,
                                 local_row, local_rows, my_rank, num_ranks, &rf, stream);
}

static int combine_reduce_launch(const void *slots, const void *topk_idx, int idx_is_i32, const float *topk_weights,
                                 const int32_t *send_data_offset, const int32_t *send_token_idx_small, int T, int K,
                                 int H, int E, void *out, const uint64_t *epoch_ctr, size_t parity_stride, const void *x_local,
                                 const int32_t *local_row, int local_rows, int my_rank, int num_ranks, const ReduceFlags *rf, void *stream)
{
    if ((send_data_offset == nullptr) != (send_token_idx_small == nullptr)) return MI_EP_EINVAL;
    if (x_local && (!local_row || local_rows <= 0 || num_ranks <= 0 || E % num_ranks || my_rank < 0 || my_rank >= num_ranks ||
                    send_data_offset))
        return MI_EP_EINVAL;
    if (T < 0 || K <= 0 || K > MI_EP_MAX_TOPK || H <= 0 || H % 8 || E <= 0) return MI_EP_EINVAL;
    if (T == 0 && !rf) return MI_EP_OK;           // (flagged form: an empty batch still completes the call counter -- one idle workgroup)
    if (T > 0 && (!slots || !topk_idx || !out)) return MI_EP_EINVAL;
    const int nchunks = H / 8;
    const int max_segs = (nchunks + kWave - 1) / kWave;
    // one wave per (token, 512-element segment) at every size: measured at C2 (4096 tokens) 95.8 us against 103 us with one wave
    // walking the whole 14 KB row (tools/probes/reduce_sweep.sh) -- more, shorter waves keep more loads in flight per CU
    int segs = 1;
    static const long long target = getenv("MI_EP_REDUCE_WAVES") ? atoll(getenv("MI_EP_REDUCE_WAVES")) : (1ll << 40);
    while (segs < max_segs && (long long)T * segs < target) segs <<= 1;
    if (segs > max_segs) segs = max_segs;
    // flagged form: the waves WAIT for rows -- keep the launch to ~1024 workgroups (half the chip's slots), so that a chip shared by several
    // processes (the one-GPU test setups) always has room for the launches that produce those rows
    if (rf) while (segs > 1 && (long long)T * segs > 4ll * (rf->max_blocks > 0 ? rf->max_blocks : 1024)) segs >>= 1;
    const long long waves = (long long)T * segs;
    if (waves > 0x7fffffffll) return MI_EP_EINVAL;          // the kernel's wave index is 32 bits wide
    const int wpb = 4;
    const long long blocks = std::max<long long>((waves + wpb - 1) / wpb, 1);
    hipStream_t s = (hipStream_t)stream;
    const Parity par = make_parity(epoch_ctr, 0, parity_stride);
    const ReduceFlags rfv = rf ? *rf : ReduceFlags{};
    // top-k <= 8 (DeepSeek-V3) gets its own instantiation: half the row registers, twice the waves per SIMD
#define MI_EP_REDUCE(I32, KMAX, FL)                                                                                                \
    combine_reduce_kernel<I32, KMAX, FL><<<(int)blocks, kWave * wpb, 0, s>>>((const uint8_t *)slots, mi_ep_combine_row_bytes(H), topk_idx, \
                                                                         topk_weights, send_data_offset, send_token_idx_small, T, K, H,  \
                                                                         E, segs, (uint16_t *)out, par, (const uint8_t *)x_local,   \
                                                                         local_row, local_rows, my_rank, x_local ? E / num_ranks : 1, rfv)
#define MI_EP_REDUCE_K(FL)                                                                                                          \
    do {                                                                                                                            \
        if (K <= 8) { if (idx_is_i32) MI_EP_REDUCE(true, 8, FL); else MI_EP_REDUCE(false, 8, FL); }                                 \
        else { if (idx_is_i32) MI_EP_REDUCE(true, MI_EP_MAX_TOPK, FL); else MI_EP_REDUCE(false, MI_EP_MAX_TOPK, FL); }              \
    } while (0)
    if (rf) MI_EP_REDUCE_K(true); else MI_EP_REDUCE_K(false);
#undef MI_EP_REDUCE_K
#undef MI_EP_REDUCE
    return launch_status();
}
