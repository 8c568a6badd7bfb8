// Window flag protocol + the notify (counts all-gather) exchange for gfx950.
// Replaces the reference's 32-byte float flag lines and magic-tagged notify flags
// (csrc/deepep/ops/op_kernel/cam_moe_dispatch_normal.h:496-502,584-631; notify_dispatch.h:247-338).
//
// MI355X design: flags are 8-byte monotonically increasing epochs (no clear pass, no ping-pong of the
// flag words); notify values travel as 8-byte {epoch, value} granules so the data is its own flag.
// Only these 8-byte words are touched with system-scope atomics; bulk payload hand-off is ordered by
// kernel boundaries (post kernel -> signal kernel | wait kernel -> consume kernel).
#include "device_once.h"
#include "ep_common.h"

namespace mi_ep {

__global__ void signal_kernel(PeerPtrs peers, int W, int my_rank, uint64_t epoch)
{
    const int d = threadIdx.x;
    if (d < W) sys_store_u64((uint64_t *)peers.p[d] + my_rank, epoch);
}

__global__ void wait_kernel(const uint64_t *__restrict__ flags, int W, uint64_t epoch, int32_t *status,
                            uint64_t timeout_ticks)
{
    const int s = threadIdx.x;
    if (s >= W) return;
    const uint64_t t0 = ticks_100mhz();
    while (sys_load_u64(flags + s) < epoch) {
        __builtin_amdgcn_s_sleep(8);
        if (ticks_100mhz() - t0 > timeout_ticks) {
            report_status(status, 1 + s);
            return;
        }
    }
}

__global__ void notify_post_kernel(PeerPtrs peers, int W, int my_rank, int E, const int32_t *__restrict__ cnt,
                                   int num_tokens, uint32_t epoch, PeerPtrs sig_peers, uint64_t sig_epoch)
{
    // grid.x = W destinations, threads sweep the E+1 values; sig_epoch != 0 also raises this rank's "rows staged" flag
    // at the destination (the fused form of notify_post + signal: both only need the preceding kernel boundary)
    const int d = blockIdx.x;
    uint64_t *row = (uint64_t *)peers.p[d] + (size_t)my_rank * (E + 1);
    for (int e = threadIdx.x; e <= E; e += blockDim.x) {
        const uint32_t v = (e < E) ? (uint32_t)cnt[e] : (uint32_t)num_tokens;
        sys_store_u64_relaxed(row + e, ((uint64_t)epoch << 32) | v);
    }
    if (sig_epoch && threadIdx.x == 0) sys_store_u64((uint64_t *)sig_peers.p[d] + my_rank, sig_epoch);
}

// signal + wait in one launch (combine: rows pushed -> tell every owner, then wait for every expert rank)
__global__ void signal_wait_kernel(PeerPtrs peers, const uint64_t *__restrict__ flags, int W, int my_rank, EpochRef er,
                                   uint64_t *epoch_bump, int32_t *status, uint64_t timeout_ticks)
{
    const int s = threadIdx.x;
    const uint64_t epoch = epoch_of(er);
    if (s < W) {
        sys_store_u64((uint64_t *)peers.p[s] + my_rank, epoch);
        const uint64_t t0 = ticks_100mhz();
        while (sys_load_u64(flags + s) < epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                report_status(status, 1 + s);
                break;
            }
        }
    }
    // one wave: every lane has read the counter before lane 0 moves it (wave-level program order)
    if (epoch_bump && s == 0) *epoch_bump = epoch;
}

__global__ void notify_wait_kernel(const uint64_t *__restrict__ notify, int n, uint32_t epoch,
                                   int32_t *__restrict__ cnt_matrix, int32_t *status, uint64_t timeout_ticks)
{
    const uint64_t t0 = ticks_100mhz();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t g;
        while (((g = sys_load_u64(notify + i)) >> 32) != epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                report_status(status, 1000 + i);
                g = 0;
                break;
            }
        }
        cnt_matrix[i] = (int32_t)(uint32_t)g;
    }
}

// One workgroup; L*W <= 2048 entries.  See mi_ep.h for the table definitions
// (reference notify_dispatch.h:386-407,434-450,473-482,553-577,606-615,665-669,715-721,759-780).
// Everything the serial reference core loops over is staged in LDS first (one coalesced pass over the counts), the
// per-source sender prefixes are wave reductions, the short dependent scans run out of LDS: ~3 us instead of ~28 us.
__device__ __forceinline__ int32_t wave_incl_scan(int32_t v, int) { return wave_incl_scan_i32(v); }

__device__ void notify_tables_body(
    const int32_t *__restrict__ cnt /*[W][E+1]*/, int W, int E, int me, int relative_pull,
    int32_t *__restrict__ recv_count, int32_t *__restrict__ recv_offset, int32_t *__restrict__ recv_tokens_per_expert,
    int32_t *__restrict__ expert_global_offset, int32_t *__restrict__ srcrank_in_expert_offset,
    int32_t *__restrict__ r_in_srcrank_offset, int32_t *__restrict__ total_recv_token, int32_t *__restrict__ max_bs,
    int32_t *__restrict__ pull_offset, int32_t *summary_host, int32_t *sm)
{
    const int L = E / W;
    const int LW = L * W;
    int32_t *c = sm;                 // [L*W] counts in idx-i order
    int32_t *pre = sm + LW;          // [W] sender prefix at my first expert
    int32_t *ego = pre + W;          // [L+1]
    int32_t *sie = ego + L + 1;      // [L*W] exclusive scan over src inside one local expert
    int32_t *mbs = sie + LW;         // [1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    // counts of my experts, idx-i order (coalesced over le for a fixed src)
    for (int i = tid; i < LW; i += blockDim.x) {
        const int le = i / W, src = i % W;
        c[i] = cnt[(size_t)src * (E + 1) + me * L + le];
    }
    // sender-side exclusive prefix up to my first expert: one wave per source rank
    for (int src = wave; src < W; src += nwaves) {
        const int32_t *row = cnt + (size_t)src * (E + 1);
        int32_t s = 0;
        for (int e = lane; e < me * L; e += 64) s += row[e];
        s = wave_sum_i32(s);
        if (lane == 0) pre[src] = s;
    }
    if (wave == 0) {
        int32_t mb = 0;
        for (int src = lane; src < W; src += 64) mb = max(mb, cnt[(size_t)src * (E + 1) + E]);
        mb = wave_max_i32(mb);
        if (lane == 0) mbs[0] = mb;
    }
    __syncthreads();
    // per source: running sender offset over my experts -- one wave per source, 64 experts per step
    for (int src = wave; src < W; src += nwaves) {
        int32_t carry = pre[src];
        for (int le0 = 0; le0 < L; le0 += 64) {
            const int le = le0 + lane;
            const int32_t v = le < L ? c[le * W + src] : 0;
            const int32_t inc = wave_incl_scan(v, lane);
            if (le < L) {
                const int32_t run = carry + inc - v;
                recv_offset[le * W + src] = run;
                pull_offset[le * W + src] = relative_pull ? run - pre[src] : run;
            }
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
    }
    // per local expert: scan over sources (W <= 64 values: serial per thread, experts in parallel)
    for (int le = tid; le < L; le += blockDim.x) {
        int32_t s = 0;
        for (int src = 0; src < W; ++src) {
            sie[le * W + src] = s;
            s += c[le * W + src];
        }
        ego[le] = s;
        recv_tokens_per_expert[le] = s;
    }
    __syncthreads();
    if (wave == 0) {                 // exclusive scan of the per-expert totals, 64 per step
        int32_t carry = 0;
        for (int le0 = 0; le0 < L; le0 += 64) {
            const int le = le0 + lane;
            const int32_t v = le < L ? ego[le] : 0;
            const int32_t inc = wave_incl_scan(v, lane);
            if (le < L) ego[le] = carry + inc - v;
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
        if (lane == 0) ego[L] = carry;
    }
    __syncthreads();
    if (summary_host) {
        // The host's words go out first (they cross PCIe while the tables below are stored), every word exactly once, >= 0, in NO
        // particular order and with no fence: the host pre-sets every word it reads to -1 and waits for each.  (Per-expert counts, a
        // barrier, a system fence and a release store of the total -- the ordered form -- kept the workgroup waiting for two PCIe
        // write round trips: ~3 us of the 4.8 us this phase took.)
        for (int le = tid; le < L; le += blockDim.x)
            __hip_atomic_store(summary_host + 2 + le, ego[le + 1] - ego[le], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (tid == 0) {
            __hip_atomic_store(summary_host + 1, mbs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(summary_host + 0, ego[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for (int i = tid; i < LW; i += blockDim.x) {
        const int le = i / W;
        srcrank_in_expert_offset[i] = sie[i];
        r_in_srcrank_offset[i] = 0;
        recv_count[i] = ego[le] + sie[i] + c[i];
    }
    for (int le = tid; le < L; le += blockDim.x) expert_global_offset[le] = ego[le];
    if (tid == 0) {
        total_recv_token[0] = ego[L];
        max_bs[0] = mbs[0];
    }
}

__global__ __launch_bounds__(256) void notify_tables_kernel(
    const int32_t *__restrict__ cnt /*[W][E+1]*/, int W, int E, int me, int relative_pull,
    int32_t *__restrict__ recv_count, int32_t *__restrict__ recv_offset, int32_t *__restrict__ recv_tokens_per_expert,
    int32_t *__restrict__ expert_global_offset, int32_t *__restrict__ srcrank_in_expert_offset,
    int32_t *__restrict__ r_in_srcrank_offset, int32_t *__restrict__ total_recv_token, int32_t *__restrict__ max_bs,
    int32_t *__restrict__ pull_offset, int32_t *summary_host)
{
    extern __shared__ __attribute__((aligned(16))) int32_t sm[];
    notify_tables_body(cnt, W, E, me, relative_pull, recv_count, recv_offset, recv_tokens_per_expert, expert_global_offset,
                       srcrank_in_expert_offset, r_in_srcrank_offset, total_recv_token, max_bs, pull_offset, summary_host, sm);
}

// Optional sender half of the exchange, run by the same workgroup before it starts to wait (mi_ep_notify_exchange_tables):
// this rank's E+1 count granules and its "rows staged" flag to every peer.
struct NotifyPost {
    PeerPtrs notify, flags;
    const int32_t *cnt;
    int num_tokens;
    uint64_t sig_epoch;      // 0 = nothing to post
};

// notify_wait + wait + notify_tables in one launch: the workgroup first collects the W*(E+1) count granules and the W
// "rows staged" flags of this call (bounded spins), then derives the tables from the counts it just wrote.
__global__ __launch_bounds__(1024) void notify_wait_tables_kernel(NotifyPost post,
    const uint64_t *__restrict__ notify_base, uint32_t notify_epoch_in, const uint64_t *__restrict__ flags, uint64_t flag_epoch_in,
    const uint64_t *epoch_ctr, uint64_t *epoch_bump, size_t notify_parity_stride,
    int32_t *__restrict__ cnt, int W, int E, int me, int relative_pull, int32_t *__restrict__ recv_count,
    int32_t *__restrict__ recv_offset, int32_t *__restrict__ recv_tokens_per_expert, int32_t *__restrict__ expert_global_offset,
    int32_t *__restrict__ srcrank_in_expert_offset, int32_t *__restrict__ r_in_srcrank_offset,
    int32_t *__restrict__ total_recv_token, int32_t *__restrict__ max_bs, int32_t *__restrict__ pull_offset,
    int32_t *summary_host, int32_t *status, uint64_t timeout_ticks, int32_t *__restrict__ wait_cost_stats)
{
    extern __shared__ __attribute__((aligned(16))) int32_t sm[];
    const int n = W * (E + 1);
#ifdef NOTIFY_TIMING
    uint64_t tk[8]; int ntk = 0;
#define NT_TICK() tk[ntk++] = wall_clock64();
#else
#define NT_TICK()
#endif
    NT_TICK()
    // this thread's first granule value is requested together with the call counter below: two independent loads of words the previous
    // kernels wrote, one round trip instead of two (the post phase took 3.2-3.8 us of the kernel's 11)
    int32_t *sc = sm;                                     // [n] the counts as they arrive: the tables are derived from LDS, not read back
    int32_t *sm_tables = sm + ((n + 3) & ~3);
    uint32_t pv0 = 0;
    if (post.sig_epoch && (int)threadIdx.x < n) {
        const int e0 = (int)threadIdx.x % (E + 1);
        pv0 = (e0 < E) ? (uint32_t)post.cnt[e0] : (uint32_t)post.num_tokens;
    }
    // device-resident epoch (graph-replayable calls): this call = counter + 1, and the notify granules ping-pong by its parity
    const uint64_t ep64 = epoch_ctr ? *epoch_ctr + 1 : 0;
    const uint32_t notify_epoch = epoch_ctr ? (uint32_t)ep64 : notify_epoch_in;
    const uint64_t flag_epoch = epoch_ctr ? ep64 : flag_epoch_in;
    const size_t npoff = epoch_ctr ? (size_t)(ep64 & 1ull) * notify_parity_stride : 0;
    const uint64_t *notify = (const uint64_t *)((const uint8_t *)notify_base + npoff);
    if (post.sig_epoch) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int d = i / (E + 1), e = i - d * (E + 1);
            const uint32_t v = i < (int)blockDim.x ? pv0 : ((e < E) ? (uint32_t)post.cnt[e] : (uint32_t)post.num_tokens);
            sys_store_u64_relaxed((uint64_t *)((uint8_t *)post.notify.p[d] + npoff) + (size_t)me * (E + 1) + e, ((uint64_t)notify_epoch << 32) | v);
        }
        if (threadIdx.x < W) sys_store_u64((uint64_t *)post.flags.p[threadIdx.x] + me, flag_epoch);
    }
    NT_TICK()
    const uint64_t t0 = ticks_100mhz();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        uint64_t g;
        while (((g = sys_load_u64(notify + i)) >> 32) != notify_epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                report_status(status, 1000 + i);
                g = 0;
                break;
            }
        }
        cnt[i] = (int32_t)(uint32_t)g;                    // (for the kernels that follow)
        sc[i] = (int32_t)(uint32_t)g;
    }
    if (flags && threadIdx.x < W) {
        while (sys_load_u64(flags + threadIdx.x) < flag_epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                report_status(status, 1 + threadIdx.x);
                break;
            }
        }
        // diagnose: microseconds this rank waited for source rank `threadIdx.x`'s rows, accumulated across calls
        // (reference cam_moe_dispatch_normal.h:580-602)
        if (wait_cost_stats) atomicAdd(wait_cost_stats + threadIdx.x, (int32_t)((ticks_100mhz() - t0) / 100));
    }
    NT_TICK()
    __syncthreads();
    NT_TICK()
    // every thread has read the counter (before the barrier above); later kernels of this call read it with add = 0
    if (epoch_bump && threadIdx.x == 0) *epoch_bump = ep64;
    notify_tables_body(sc, W, E, me, relative_pull, recv_count, recv_offset, recv_tokens_per_expert, expert_global_offset,
                       srcrank_in_expert_offset, r_in_srcrank_offset, total_recv_token, max_bs, pull_offset, summary_host, sm_tables);
#ifdef NOTIFY_TIMING
    NT_TICK()
    if (threadIdx.x == 0) for (int i = 0; i < ntk; ++i) ((uint64_t *)(cnt + ((n + 17) & ~1)))[i] = tk[i];
#endif
}

}  // namespace mi_ep

using namespace mi_ep;

// The count exchange keeps all W * (E + 1) counts in LDS next to the tables' scratch; a CU has 160 KB.  Shapes beyond that (W = 64 with
// E = 2048, say) are refused with MI_EP_EINVAL by the entry points -- they are far outside one xGMI node (W <= 8) -- instead of failing
// at launch; mi_ep_notify_lds_bytes lets a host check up front.
static constexpr size_t kNotifyLdsMax = 160 * 1024;
extern "C" size_t mi_ep_notify_lds_bytes(int W, int E)
{
    if (W <= 0 || E <= 0 || E % W) return 0;
    const int L = E / W;
    return (size_t)(2 * L * W + W + L + 2 + ((W * (E + 1) + 3) & ~3)) * sizeof(int32_t);
}

// dynamic LDS of notify_wait_tables_kernel: the tables' scratch + the W * (E + 1) counts (84 KB at E = 2048, W = 8: above the 64 KB default)
static size_t notify_lds_bytes(int W, int E)
{
    static PerDeviceOnce attr_once;
    if (attr_once.need()) {
        (void)hipFuncSetAttribute((const void *)notify_wait_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNotifyLdsMax);
    }
    return mi_ep_notify_lds_bytes(W, E);
}

static uint64_t ms_to_ticks(int ms) { return (uint64_t)(ms > 0 ? ms : 10000) * 100000ull; }

static int fill_peers(PeerPtrs &pp, const void *const *host, int W)
{
    if (!host || W <= 0 || W > MI_EP_MAX_RANKS) return MI_EP_EINVAL;
    for (int i = 0; i < W; ++i) {
        if (!host[i]) return MI_EP_EINVAL;
        pp.p[i] = const_cast<void *>(host[i]);
    }
    return MI_EP_OK;
}

extern "C" const char *mi_ep_version(void) { return "mi_ep 0.1 gfx950"; }

extern "C" int mi_ep_signal(uint64_t *const *peer_flags_host, int W, int my_rank, uint64_t epoch, void *stream)
{
    PeerPtrs pp;
    if (fill_peers(pp, (const void *const *)peer_flags_host, W) || my_rank < 0 || my_rank >= W) return MI_EP_EINVAL;
    signal_kernel<<<1, kWave, 0, (hipStream_t)stream>>>(pp, W, my_rank, epoch);
    return launch_status();
}

extern "C" int mi_ep_wait(const uint64_t *my_flags, int W, uint64_t epoch, int32_t *status, int timeout_ms,
                          void *stream)
{
    if (!my_flags || !status || W <= 0 || W > MI_EP_MAX_RANKS) return MI_EP_EINVAL;
    wait_kernel<<<1, kWave, 0, (hipStream_t)stream>>>(my_flags, W, epoch, status, ms_to_ticks(timeout_ms));
    return launch_status();
}

extern "C" int mi_ep_notify_post(uint64_t *const *peer_notify_host, int W, int my_rank, int E,
                                 const int32_t *num_tokens_per_expert, int num_tokens, uint32_t epoch, void *stream)
{
    PeerPtrs pp;
    if (fill_peers(pp, (const void *const *)peer_notify_host, W) || my_rank < 0 || my_rank >= W || E <= 0 ||
        !num_tokens_per_expert || epoch == 0)
        return MI_EP_EINVAL;
    notify_post_kernel<<<W, 256, 0, (hipStream_t)stream>>>(pp, W, my_rank, E, num_tokens_per_expert, num_tokens, epoch, pp, 0);
    return launch_status();
}

extern "C" int mi_ep_notify_post_signal(uint64_t *const *peer_notify_host, uint64_t *const *peer_flags_host, int W, int my_rank,
                                        int E, const int32_t *num_tokens_per_expert, int num_tokens, uint32_t notify_epoch,
                                        uint64_t signal_epoch, void *stream)
{
    PeerPtrs pp, sp;
    if (fill_peers(pp, (const void *const *)peer_notify_host, W) || fill_peers(sp, (const void *const *)peer_flags_host, W) ||
        my_rank < 0 || my_rank >= W || E <= 0 || !num_tokens_per_expert || notify_epoch == 0 || signal_epoch == 0)
        return MI_EP_EINVAL;
    notify_post_kernel<<<W, 256, 0, (hipStream_t)stream>>>(pp, W, my_rank, E, num_tokens_per_expert, num_tokens, notify_epoch, sp,
                                                           signal_epoch);
    return launch_status();
}

extern "C" int mi_ep_signal_wait(uint64_t *const *peer_flags_host, const uint64_t *my_flags, int W, int my_rank, uint64_t epoch,
                                 uint64_t *epoch_ctr, int32_t *status, int timeout_ms, void *stream)
{
    PeerPtrs pp;
    if (fill_peers(pp, (const void *const *)peer_flags_host, W) || !my_flags || !status || my_rank < 0 || my_rank >= W)
        return MI_EP_EINVAL;
    const EpochRef er = epoch_ctr ? EpochRef{epoch_ctr, 1} : EpochRef{nullptr, epoch};
    signal_wait_kernel<<<1, kWave, 0, (hipStream_t)stream>>>(pp, my_flags, W, my_rank, er, epoch_ctr, status, ms_to_ticks(timeout_ms));
    return launch_status();
}

extern "C" int mi_ep_notify_wait(const uint64_t *my_notify, int W, int E, uint32_t epoch, int32_t *cnt_matrix,
                                 int32_t *status, int timeout_ms, void *stream)
{
    if (!my_notify || !cnt_matrix || !status || W <= 0 || E <= 0 || epoch == 0) return MI_EP_EINVAL;
    const int n = W * (E + 1);
    const int blocks = (n + 255) / 256 < 8 ? (n + 255) / 256 : 8;
    notify_wait_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(my_notify, n, epoch, cnt_matrix, status,
                                                               ms_to_ticks(timeout_ms));
    return launch_status();
}

extern "C" int mi_ep_notify_tables(const int32_t *cnt_matrix, int W, int E, int my_rank, int relative_pull,
                                   int32_t *recv_count, int32_t *recv_offset, int32_t *recv_tokens_per_expert,
                                   int32_t *expert_global_offset, int32_t *srcrank_in_expert_offset,
                                   int32_t *r_in_srcrank_offset, int32_t *total_recv_token, int32_t *max_bs,
                                   int32_t *pull_offset, int32_t *summary_host, void *stream)
{
    if (!cnt_matrix || W <= 0 || W > MI_EP_MAX_RANKS || E <= 0 || E % W || E > 2048 || my_rank < 0 || my_rank >= W)
        return MI_EP_EINVAL;
    const int L = E / W;
    const size_t lds = (size_t)(2 * L * W + W + L + 2) * sizeof(int32_t);
    notify_tables_kernel<<<1, 256, lds, (hipStream_t)stream>>>(cnt_matrix, W, E, my_rank, relative_pull, recv_count,
                                                              recv_offset, recv_tokens_per_expert, expert_global_offset,
                                                              srcrank_in_expert_offset, r_in_srcrank_offset,
                                                              total_recv_token, max_bs, pull_offset, summary_host);
    return launch_status();
}

extern "C" int mi_ep_notify_wait_tables(const uint64_t *my_notify, uint32_t notify_epoch, const uint64_t *my_flags,
                                        uint64_t flag_epoch, int32_t *cnt_matrix, int W, int E, int my_rank, int relative_pull,
                                        int32_t *recv_count, int32_t *recv_offset, int32_t *recv_tokens_per_expert,
                                        int32_t *expert_global_offset, int32_t *srcrank_in_expert_offset,
                                        int32_t *r_in_srcrank_offset, int32_t *total_recv_token, int32_t *max_bs,
                                        int32_t *pull_offset, int32_t *summary_host, int32_t *status, int timeout_ms,
                                        int32_t *wait_cost_stats, void *stream)
{
    if (!my_notify || !cnt_matrix || !status || notify_epoch == 0 || W <= 0 || W > MI_EP_MAX_RANKS || E <= 0 || E % W || E > 2048 ||
        my_rank < 0 || my_rank >= W)
        return MI_EP_EINVAL;
    const size_t lds = notify_lds_bytes(W, E);
    if (lds > kNotifyLdsMax) return MI_EP_EINVAL;
    NotifyPost none{};
    notify_wait_tables_kernel<<<1, 1024, lds, (hipStream_t)stream>>>(
        none, my_notify, notify_epoch, my_flags, flag_epoch, nullptr, nullptr, 0, cnt_matrix, W, E, my_rank, relative_pull, recv_count, recv_offset,
        recv_tokens_per_expert, expert_global_offset, srcrank_in_expert_offset, r_in_srcrank_offset, total_recv_token, max_bs,
        pull_offset, summary_host, status, ms_to_ticks(timeout_ms), wait_cost_stats);
    return launch_status();
}

extern "C" int mi_ep_notify_exchange_tables(uint64_t *const *peer_notify_host, uint64_t *const *peer_flags_host,
                                            const int32_t *num_tokens_per_expert, int num_tokens, const uint64_t *my_notify,
                                            uint32_t notify_epoch, const uint64_t *my_flags, uint64_t flag_epoch,
                                            int32_t *cnt_matrix, int W, int E, int my_rank, int relative_pull, int32_t *recv_count,
                                            int32_t *recv_offset, int32_t *recv_tokens_per_expert, int32_t *expert_global_offset,
                                            int32_t *srcrank_in_expert_offset, int32_t *r_in_srcrank_offset,
                                            int32_t *total_recv_token, int32_t *max_bs, int32_t *pull_offset,
                                            int32_t *summary_host, uint64_t *epoch_ctr, size_t notify_parity_stride, int32_t *status,
                                            int timeout_ms, int32_t *wait_cost_stats, void *stream)
{
    if (!my_notify || !my_flags || !cnt_matrix || !status || !num_tokens_per_expert || ((notify_epoch == 0 || flag_epoch == 0) && !epoch_ctr) ||
        W <= 0 || W > MI_EP_MAX_RANKS || E <= 0 || E % W || E > 2048 || my_rank < 0 || my_rank >= W)
        return MI_EP_EINVAL;
    NotifyPost post{};
    if (fill_peers(post.notify, (const void *const *)peer_notify_host, W) || fill_peers(post.flags, (const void *const *)peer_flags_host, W))
        return MI_EP_EINVAL;
    post.cnt = num_tokens_per_expert;
    post.num_tokens = num_tokens;
    post.sig_epoch = epoch_ctr ? 1 : flag_epoch;      // non-zero = post; the value comes from the counter when it is device-resident
    const size_t lds = notify_lds_bytes(W, E);
    if (lds > kNotifyLdsMax) return MI_EP_EINVAL;
    notify_wait_tables_kernel<<<1, 1024, lds, (hipStream_t)stream>>>(
        post, my_notify, notify_epoch, my_flags, flag_epoch, epoch_ctr, epoch_ctr, notify_parity_stride, cnt_matrix, W, E, my_rank, relative_pull, recv_count, recv_offset,
        recv_tokens_per_expert, expert_global_offset, srcrank_in_expert_offset, r_in_srcrank_offset, total_recv_token, max_bs,
        pull_offset, summary_host, status, ms_to_ticks(timeout_ms), wait_cost_stats);
    return launch_status();
}

// ---- start-up self-test of the mapped windows -----------------------------------------------------------------------
namespace mi_ep {
constexpr int kSelfTestWords = 1024;      // one 4 KiB row per (source, destination) pair
__device__ __forceinline__ uint32_t selftest_word(uint32_t tag, int src, int dst, int i)
{
    uint32_t h = tag * 0x9E3779B1u ^ (uint32_t)(src * 977 + dst * 131 + 7) * 0x85EBCA6Bu ^ (uint32_t)i * 0xC2B2AE35u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h;
}
// Round protocol (two or more rounds on the SAME addresses, a fresh tag each): a line that a rank cached while checking round r and
// that a peer rewrote in round r + 1 must not be served stale -- the one failure a single round with fresh addresses and
// cache-bypassing loads cannot show.  All checks therefore use ORDINARY loads (what the bulk kernels use), in a launch of their own.
//   gate   (round > first): wait until every peer has finished CHECKING the previous round (ack flags), so nobody's rows are
//          rewritten under a reader
//   post   block d: this rank's pattern row -> rank d's window, slot my_rank (ordinary 16-byte stores), plus one {epoch, value}
//          granule (relaxed system-scope store, the notify / low-latency count path)
//   check  one workgroup: raise "written" on every peer, wait for every peer; verify the rows peers wrote into MY window
//          (remote-write path: push transports), my row read back from every PEER (remote-read path: pull transport) and the granules;
//          then raise "checked" on every peer.
// status[0]: 0 ok, 1 + s timeout on rank s, 3000 + s bad row from s, 4000 + d bad read-back from d, 5000 + s bad granule from s.
__global__ void selftest_gate_kernel(const uint64_t *__restrict__ my_acks, int W, uint64_t prev_epoch, int32_t *status, uint64_t timeout_ticks)
{
    if ((int)threadIdx.x >= W) return;
    const uint64_t t0 = ticks_100mhz();
    while (sys_load_u64(my_acks + threadIdx.x) < prev_epoch) {
        __builtin_amdgcn_s_sleep(8);
        if (ticks_100mhz() - t0 > timeout_ticks) {
            report_status(status, 1 + threadIdx.x);
            break;
        }
    }
}
__global__ void selftest_post_kernel(PeerPtrs rows, int W, int my_rank, uint32_t tag, uint64_t epoch)
{
    const int d = blockIdx.x;
    uint32_t *dst = (uint32_t *)((uint8_t *)rows.p[d] + (size_t)my_rank * kSelfTestWords * 4);
    for (int i = threadIdx.x * 4; i < kSelfTestWords; i += blockDim.x * 4)
        *(u32x4 *)(dst + i) = u32x4{selftest_word(tag, my_rank, d, i), selftest_word(tag, my_rank, d, i + 1),
                                    selftest_word(tag, my_rank, d, i + 2), selftest_word(tag, my_rank, d, i + 3)};
    if (threadIdx.x == 0) {
        uint64_t *gran = (uint64_t *)((uint8_t *)rows.p[d] + (size_t)W * kSelfTestWords * 4) + my_rank;
        sys_store_u64_relaxed(gran, (epoch << 32) | selftest_word(tag, my_rank, d, kSelfTestWords));
    }
}
__global__ __launch_bounds__(256) void selftest_check_kernel(PeerPtrs rows, PeerPtrs flags, PeerPtrs acks, const uint64_t *__restrict__ my_flags,
                                                            int W, int my_rank, uint64_t epoch, uint32_t tag, int32_t *status,
                                                            uint64_t timeout_ticks)
{
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if ((int)threadIdx.x < W) {
        sys_store_u64((uint64_t *)flags.p[threadIdx.x] + my_rank, epoch);
        const uint64_t t0 = ticks_100mhz();
        while (sys_load_u64(my_flags + threadIdx.x) < epoch) {
            __builtin_amdgcn_s_sleep(8);
            if (ticks_100mhz() - t0 > timeout_ticks) {
                report_status(status, 1 + threadIdx.x);
                bad = 1;
                break;
            }
        }
    }
    __threadfence_system();
    __syncthreads();
    if (bad) return;
    for (int s = 0; s < W; ++s) {
        const uint32_t *mine = (const uint32_t *)((const uint8_t *)rows.p[my_rank] + (size_t)s * kSelfTestWords * 4);
        const uint32_t *theirs = (const uint32_t *)((const uint8_t *)rows.p[s] + (size_t)my_rank * kSelfTestWords * 4);
        for (int i = threadIdx.x; i < kSelfTestWords; i += blockDim.x) {
            if (mine[i] != selftest_word(tag, s, my_rank, i)) report_status(status, 3000 + s);
            if (theirs[i] != selftest_word(tag, my_rank, s, i)) report_status(status, 4000 + s);
        }
    }
    if ((int)threadIdx.x < W) {
        const int s = threadIdx.x;
        const uint64_t *gran = (const uint64_t *)((const uint8_t *)rows.p[my_rank] + (size_t)W * kSelfTestWords * 4) + s;
        // the granule travelled as a relaxed store posted BEFORE the peer's "written" flag (a release): it must be there by now
        const uint64_t g = sys_load_u64(gran);
        if ((g >> 32) != epoch || (uint32_t)g != selftest_word(tag, s, my_rank, kSelfTestWords)) report_status(status, 5000 + s);
    }
    __syncthreads();                                   // every thread's reads are done before anybody may rewrite these rows
    if ((int)threadIdx.x < W) sys_store_u64((uint64_t *)acks.p[threadIdx.x] + my_rank, epoch);
}
}  // namespace mi_ep

extern "C" size_t mi_ep_selftest_bytes(int num_ranks) { return (size_t)num_ranks * (mi_ep::kSelfTestWords * 4 + 8); }

extern "C" int mi_ep_selftest(void *const *peer_rows_host, uint64_t *const *peer_flags_host, const uint64_t *my_flags,
                              uint64_t *const *peer_acks_host, const uint64_t *my_acks, int W, int my_rank, uint64_t first_epoch,
                              int rounds, uint32_t tag, int32_t *status, int timeout_ms, void *stream)
{
    PeerPtrs rp, fp, ap;
    if (fill_peers(rp, (const void *const *)peer_rows_host, W) || fill_peers(fp, (const void *const *)peer_flags_host, W) ||
        fill_peers(ap, (const void *const *)peer_acks_host, W) || !my_flags || !my_acks || !status || my_rank < 0 || my_rank >= W ||
        first_epoch == 0 || rounds < 1 || rounds > 16)
        return MI_EP_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t ticks = ms_to_ticks(timeout_ms);
    for (int r = 0; r < rounds; ++r) {
        const uint64_t epoch = first_epoch + (uint64_t)r;
        const uint32_t rtag = tag + (uint32_t)r * 0x01000193u;
        // the acks of the epoch BEFORE the first one of this call were raised by the previous call's last round (or are 0 = nothing to wait for)
        if (epoch > 1) mi_ep::selftest_gate_kernel<<<1, 64, 0, st>>>(my_acks, W, epoch - 1, status, ticks);
        mi_ep::selftest_post_kernel<<<W, 256, 0, st>>>(rp, W, my_rank, rtag, epoch);
        mi_ep::selftest_check_kernel<<<1, 256, 0, st>>>(rp, fp, ap, my_flags, W, my_rank, epoch, rtag, status, ticks);
    }
    return launch_status();
}

// ---- start-up self-test, second leg: the IN-LAUNCH hand-off of the two-launch low-latency forms -----------------------------------------
// What mi_ep_selftest above cannot show: its checks run in a launch of their own, behind a kernel boundary.  The two-launch low-latency forms
// hand rows over INSIDE running launches (dispatch.hip: stage_*_body LATE + tag / ll_wait_pack_kernel; combine.hip: combine_push_kernel
// FLAGGED / combine_reduce_body FLAGGED; the reference's per-token flag wait, moe_distribute_combine_v2.h:952-1002,
// moe_distribute_dispatch_v2.h:1159-1171):
//   producer wave   row payload with write-through stores (sc0 sc1, 16 B per lane, buffer descriptor)  ->  s_waitcnt vmcnt(0)  ->  one lane:
//                   the tag in the row's meta word (same store flavour, "tagged" rows) or the row's flag word in the owner's control area
//                   (relaxed system-scope store, "flagged" rows)
//   consumer wave   one lane polls that word (relaxed, system scope), the wave then reads the payload with system-scope loads (ld_sys_b128)
// This leg runs exactly those instruction sequences: one launch per round whose workgroups [0, W) produce for rank d and [W, 2W) consume from
// rank s, kILRows tagged rows (128-byte aligned, 16 B of meta behind the payload) and kILRows flagged rows (16-byte aligned: neighbours share
// cache lines) per pair, rounds alternating between the two ping-pong halves so that round r + 2 rewrites the addresses of round r under a
// fresh pattern.  After checking a row the consumer reads it once more with ORDINARY loads and keeps nothing: the lines stay in its L1 / L2
// the way any earlier reader may have left them, and the next visit must not be answered from there.  A producer starts round r only when
// its consumer has acknowledged round r - 1 (nobody's rows are rewritten under a reader).
// status[0]: 1 + s gate / word never arrived from rank s, 7000 + s stale or corrupt TAGGED payload from s, 7500 + s the same for a FLAGGED row.
namespace mi_ep {
constexpr int kILRows = 8;
constexpr int kILPayload = 4096;
constexpr int kILTagStride = kILPayload + 128;
constexpr int kILFlagStride = kILPayload + 16;
constexpr size_t kILPairBytes = (size_t)kILRows * (kILTagStride + kILFlagStride);
struct InLaunchTest {
    PeerPtrs rows, flags, acks;
    const uint64_t *my_acks;
    size_t rows_half_stride, flags_half_stride;
    int W, my_rank;
    uint64_t epoch;
    uint32_t tag;
    int skip_payload;          // test hook: this producer raises tags / flags without rewriting the payload (what a stale line looks like)
};
__device__ __forceinline__ uint32_t il_tag24(uint32_t tag, uint64_t epoch) { return ((tag ^ (uint32_t)epoch * 0x9E37u) & 0xFFFFFFu) | 1u; }
__global__ __launch_bounds__(256) void selftest_inlaunch_kernel(InLaunchTest a, int32_t *status, uint64_t timeout_ticks)
{
    __shared__ int bad;
    const int W = a.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t half = (size_t)(a.epoch & 1ull);
    const uint64_t t0 = ticks_100mhz();
    if (tid == 0) bad = 0;
    __syncthreads();
    if ((int)blockIdx.x < W) {                                  // ---- producer for rank d
        const int d = blockIdx.x;
        if (tid == 0) {
            while (sys_load_u64(a.my_acks + d) < a.epoch - 1) {
                __builtin_amdgcn_s_sleep(8);
                if (ticks_100mhz() - t0 > timeout_ticks) {
                    report_status(status, 1 + d);
                    bad = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (bad) return;
        uint8_t *base = (uint8_t *)a.rows.p[d] + half * a.rows_half_stride + (size_t)a.my_rank * kILPairBytes;
        for (int row = wave; row < 2 * kILRows; row += (int)blockDim.x / kWave) {
            const bool flagged = row >= kILRows;
            uint8_t *rp = flagged ? base + (size_t)kILRows * kILTagStride + (size_t)(row - kILRows) * kILFlagStride : base + (size_t)row * kILTagStride;
            const __amdgpu_buffer_rsrc_t rs = sys_row_rsrc(rp, kILPayload + 16);
            if (!a.skip_payload) {
#pragma unroll
                for (int u = 0; u < kILPayload / (kWave * 16); ++u) {
                    const int i = (u * kWave + lane) * 4 + row * (kILPayload / 4);
                    const u32x4 v = u32x4{selftest_word(a.tag, a.my_rank, d, i), selftest_word(a.tag, a.my_rank, d, i + 1),
                                          selftest_word(a.tag, a.my_rank, d, i + 2), selftest_word(a.tag, a.my_rank, d, i + 3)};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, v), rs, (u * kWave + lane) * 16, 0, 17);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the payload is at its owner before the word below says so
            if (lane == 0) {
                if (!flagged) {
                    const u32x4 m = u32x4{a.tag, (uint32_t)row, 0u, (uint32_t)a.my_rank | (il_tag24(a.tag, a.epoch) << 8)};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_u32x4, m), rs, kILPayload, 0, 17);
                } else {
                    uint32_t *fl = (uint32_t *)((uint8_t *)a.flags.p[d] + half * a.flags_half_stride) + a.my_rank * kILRows + (row - kILRows);
                    __hip_atomic_store(fl, (uint32_t)a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        return;
    }
    // ---- consumer of rank s
    const int s = (int)blockIdx.x - W;
    const uint8_t *base = (const uint8_t *)a.rows.p[a.my_rank] + half * a.rows_half_stride + (size_t)s * kILPairBytes;
    for (int row = wave; row < 2 * kILRows; row += (int)blockDim.x / kWave) {
        const bool flagged = row >= kILRows;
        const uint8_t *rp = flagged ? base + (size_t)kILRows * kILTagStride + (size_t)(row - kILRows) * kILFlagStride : base + (size_t)row * kILTagStride;
        int late = 0;
        if (lane == 0) {
            const uint32_t *word = flagged ? (const uint32_t *)((const uint8_t *)a.flags.p[a.my_rank] + half * a.flags_half_stride) + s * kILRows + (row - kILRows)
                                           : (const uint32_t *)(rp + kILPayload) + 3;
            const uint32_t want = flagged ? (uint32_t)a.epoch : il_tag24(a.tag, a.epoch);
            for (;;) {
                const uint32_t w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if ((flagged ? w : w >> 8) == want) break;
                __builtin_amdgcn_s_sleep(2);
                if (ticks_100mhz() - t0 > timeout_ticks) {
                    report_status(status, 1 + s);
                    late = 1;
                    break;
                }
            }
        }
        asm volatile("" ::: "memory");                         // the wave reconverges behind lane 0's wait: the row is read after its word was seen
        if (__builtin_amdgcn_readfirstlane(late)) continue;
        const __amdgpu_buffer_rsrc_t rs = sys_row_rsrc(rp, kILPayload);
        u32x4 v[kILPayload / (kWave * 16)];
#pragma unroll
        for (int u = 0; u < kILPayload / (kWave * 16); ++u) v[u] = ld_sys_b128(rs, (uint32_t)(u * kWave + lane) * 16u);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < kILPayload / (kWave * 16); ++u) {
            const int i = (u * kWave + lane) * 4 + row * (kILPayload / 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) ok = ok && v[u][j] == selftest_word(a.tag, s, a.my_rank, i + j);
        }
        if (!ok) report_status(status, (flagged ? 7500 : 7000) + s);
        // leave the row's lines in this CU's L1 and this XCD's L2 as an ordinary reader would (nothing is kept from the loads)
#pragma unroll
        for (int u = 0; u < kILPayload / (kWave * 16); ++u) {
            u32x4 w = ((const u32x4 *)rp)[u * kWave + lane];
            asm volatile("" ::"v"(w));
        }
    }
    __syncthreads();                                           // every wave's reads are done before the producer may rewrite these rows
    if (tid == 0) sys_store_u64((uint64_t *)a.acks.p[s] + a.my_rank, a.epoch);
}
}  // namespace mi_ep

extern "C" size_t mi_ep_selftest_inlaunch_bytes(int num_ranks) { return (size_t)num_ranks * mi_ep::kILPairBytes; }
extern "C" size_t mi_ep_selftest_inlaunch_flag_words(int num_ranks) { return (size_t)num_ranks * mi_ep::kILRows; }

extern "C" int mi_ep_selftest_inlaunch(void *const *peer_rows_host, size_t rows_half_stride, uint32_t *const *peer_row_flags_host,
                                       size_t flags_half_stride, uint64_t *const *peer_acks_host, const uint64_t *my_acks, int W, int my_rank,
                                       uint64_t first_epoch, int rounds, uint32_t tag, int skip_payload_from_round, int32_t *status,
                                       int timeout_ms, void *stream)
{
    mi_ep::InLaunchTest a{};
    if (fill_peers(a.rows, (const void *const *)peer_rows_host, W) || fill_peers(a.flags, (const void *const *)peer_row_flags_host, W) ||
        fill_peers(a.acks, (const void *const *)peer_acks_host, W) || !my_acks || !status || my_rank < 0 || my_rank >= W || first_epoch == 0 ||
        rounds < 1 || rounds > 16 || rows_half_stride < mi_ep_selftest_inlaunch_bytes(W) ||
        flags_half_stride < mi_ep_selftest_inlaunch_flag_words(W) * 4)
        return MI_EP_EINVAL;
    a.my_acks = my_acks, a.rows_half_stride = rows_half_stride, a.flags_half_stride = flags_half_stride, a.W = W, a.my_rank = my_rank;
    const uint64_t ticks = ms_to_ticks(timeout_ms);
    for (int r = 0; r < rounds; ++r) {
        a.epoch = first_epoch + (uint64_t)r;
        a.tag = tag + (uint32_t)r * 0x01000193u;
        a.skip_payload = skip_payload_from_round >= 0 && r >= skip_payload_from_round;
        mi_ep::selftest_inlaunch_kernel<<<2 * W, 256, 0, (hipStream_t)stream>>>(a, status, ticks);
    }
    return launch_status();
}

// ---- diagnose helpers (only launched when the caller passes a stats tensor) -------------------------------------------
namespace mi_ep {
__global__ void timestamp_kernel(uint64_t *dst) { *dst = ticks_100mhz(); }
__global__ void elapsed_add_kernel(int32_t *stats, int n, const uint64_t *t_start)
{
    const int32_t us = (int32_t)((ticks_100mhz() - *t_start) / 100);
    for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(stats + i, us);
}
}  // namespace mi_ep

extern "C" int mi_ep_timestamp(uint64_t *dst, void *stream)
{
    if (!dst) return MI_EP_EINVAL;
    mi_ep::timestamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(dst);
    return launch_status();
}

extern "C" int mi_ep_elapsed_add(int32_t *stats, int n, const uint64_t *t_start, void *stream)
{
    if (!stats || !t_start || n <= 0) return MI_EP_EINVAL;
    mi_ep::elapsed_add_kernel<<<1, 64, 0, (hipStream_t)stream>>>(stats, n, t_start);
    return launch_status();
}
