// MI355X host runtime of deep_ep (see deep_ep.hpp).  Behaviour follows the reference host runtime
// csrc/deepep/deep_ep.cpp (cited per function); the implementation is HIP-native:
//   * one fine-grained device allocation per rank is the symmetric window, mapped into every peer with hipIpc;
//   * every op is a short chain of launches on the CALLER'S current stream (post -> signal | wait -> consume);
//   * the one unavoidable host sync of normal dispatch (output size) is a spin on a pinned host word the
//     notify kernel writes, instead of the reference's two .item() calls + a .to(CPU).
#include "deep_ep.hpp"

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include "mi_ep.h"

namespace deep_ep {

namespace {
constexpr int64_t kCtrlBytes = 4ll << 20;          // flags + notify granules + LL count granules
constexpr int kFlagGroupSlots = 64;                // MI_EP_MAX_RANKS
constexpr int64_t kOffFlags = 0;                   // 8 groups x 64 x u64
constexpr int64_t kOffEpochs = 8 << 10;                  // u64 completed-call counters, one per family (device-resident epochs)
constexpr int64_t kOffNotify = 64 << 10;           // 2 parities x W x (E+1) u64  (<= 2 x 64 x 2049 x 8 = 2.1 MB)
constexpr int64_t kNotifyParityBytes = 1100 << 10;
constexpr int64_t kOffLLCounts = kOffNotify + 2 * kNotifyParityBytes;   // 2 parities x 2048 u64
constexpr int64_t kLLCountsParityBytes = 2048 * 8;
// row flags of the two-launch combine (mi_ep_combine_push_flagged): 2 halves x 131072 uint32, word t * K + k of a rank's combine slots
constexpr int64_t kOffRowFlags = 2560 << 10;
constexpr int64_t kRowFlagsParityBytes = 512 << 10;
static_assert(kOffLLCounts + 2 * kLLCountsParityBytes <= kOffRowFlags && kOffRowFlags + 2 * kRowFlagsParityBytes <= kCtrlBytes, "control area layout");
enum Family { kDispatch = 0, kCombine = 1, kLLDispatch = 2 };
enum FlagGroup { kFlagDispatch = 0, kFlagCombine = 1, kFlagSelfTestAck = 6, kFlagSelfTest = 7 };
constexpr int kMaxTotalTokens = 131072;            // reference MAX_TOTAL_TOKENS (deep_ep.cpp:37)

hipStream_t cur_stream() { return c10::hip::getCurrentHIPStream().stream(); }

// id of the graph capture `st` is recording into, 0 when it is not capturing
uint64_t capture_id_of(hipStream_t st)
{
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo(st, &status, &id) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return status == hipStreamCaptureStatusActive ? (uint64_t)id + 1 : 0;
}

int quant_mode_of(bool use_quant, const std::string &quant_type)
{
    if (!use_quant) return MI_EP_QUANT_NONE;
    if (quant_type == "int8_ll") return MI_EP_QUANT_INT8_NOEPS;     // low-latency rounding (no epsilon), a2a strategy only
    // per-token FP8 E4M3 is Ascend950-only in the reference (deep_ep.cpp:338-343); gfx950 converts to OCP FP8 natively, so it is
    // served here.  The block-scaled MXFP8 / MXFP4 modes are not built.
    if (quant_type == "pertoken_fp8_e4m3") return MI_EP_QUANT_FP8_E4M3;
    EP_HOST_ASSERT_S(quant_type == "int8", quant_type, " is not supported on this device, please use int8, pertoken_fp8_e4m3 or bf16 instead.");
    return MI_EP_QUANT_INT8;
}

// dtype of a one-byte-per-element payload tensor (reference deep_ep.cpp:344-365)
at::ScalarType payload_dtype(int qm) { return qm == MI_EP_QUANT_FP8_E4M3 ? at::kFloat8_e4m3fn : at::kChar; }
}  // namespace

EventHandle::EventHandle()
{
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) hipEventRecord(ev, cur_stream());
}
void EventHandle::current_stream_wait() const
{
    if (ev) hipStreamWaitEvent(cur_stream(), ev, 0);
}

// ------------------------------------------------------------------------------------------------
Buffer::Buffer(int64_t rank, int64_t num_ranks, int64_t num_nvl_bytes, int64_t num_rdma_bytes, bool low_latency_mode,
               std::string moe_all_to_all_group_name)
    : rank(rank),
      num_ranks(num_ranks),
      num_nvl_bytes(num_nvl_bytes),
      num_rdma_bytes(num_rdma_bytes),
      low_latency_mode(low_latency_mode),
      group_name(std::move(moe_all_to_all_group_name))
{
    EP_HOST_ASSERT(0 <= rank and rank < num_ranks);
    EP_HOST_ASSERT_S(num_ranks <= MI_EP_MAX_RANKS, "one xGMI domain holds at most ", MI_EP_MAX_RANKS, " ranks");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        throw EPException("HIP Assertion", __FILE__, __LINE__,
                          "no HIP device visible: deep_ep_cpp needs an AMD GPU (there is no CPU fallback)");
    HIP_CHECK(hipGetDevice(&device_id));
    timeout_ms = get_value_from_env("DEEPEP_TIMEOUT_MS", 30000);
    // Host-runtime env knobs of the reference that change the data layout (deep_ep.cpp:62,866-874,939,1076-1079):
    //  * MOE_SHARED_EXPERT_RANK_NUM = S > 0 turns the first S ranks into shared-expert ranks (one local expert each; the others hold
    //    num_experts / (W - S)); low_latency_dispatch / low_latency_combine / fused_deep_moe follow it (shared_view below), the
    //    normal-mode ops ignore it as in the reference (deep_ep.cpp:684).  S must divide W: a shared rank's receive bound is
    //    global_bs / S rows (deep_ep.cpp:870), which only holds when every shared rank serves W / S sources.
    //  * MOE_ENABLE_TOPK_NEG_ONE=1 makes the reference pass x_active_mask = (topk_idx >= 0) so that -1 selections are
    //    skipped; without it -1 is outside the reference's contract.  Here ids < 0 (and >= num_experts) are ALWAYS skipped by
    //    every kernel (layout, stage, reduce), i.e. both settings of the knob behave like "1"; the value is validated only.
    shared_expert_rank_num = get_value_from_env("MOE_SHARED_EXPERT_RANK_NUM", 0);
    EP_HOST_ASSERT_S(shared_expert_rank_num >= 0 && shared_expert_rank_num < num_ranks &&
                         (shared_expert_rank_num == 0 || num_ranks % shared_expert_rank_num == 0),
                     "MOE_SHARED_EXPERT_RANK_NUM=", shared_expert_rank_num, " must be in [0, num_ranks) and divide num_ranks (", num_ranks, ")");
    const int enable_neg_one = get_value_from_env("MOE_ENABLE_TOPK_NEG_ONE", 0);
    EP_HOST_ASSERT_S(enable_neg_one == 0 || enable_neg_one == 1, "MOE_ENABLE_TOPK_NEG_ONE must be 0 or 1, got ", enable_neg_one);

    // Window budget.  The reference sizes its HCCL window with HCCL_BUFFSIZE (MB); DEEPEP_WINDOW_BYTES plays that
    // role here.  Default 6 GiB = six ~1-GiB regions: dispatch x2, combine x2, low-latency dispatch x2 (ping-pong),
    // enough for 8192 tok x top-8 x 7168 BF16 per region; MI355X has 288 GB.
    // The window is FOUR allocations (segments): the 4 MiB control area and one allocation per family holding both
    // ping-pong regions.  Measured on ROCm 7.2 / MI355X (tools/probes/ipc_open_time.py): hipIpcOpenMemHandle of an
    // allocation of 2 GiB or more never returns (2040 MiB maps in 0.1 s, 2056 MiB hangs), so a segment stays below that.
    long long want = get_ll_from_env("DEEPEP_WINDOW_BYTES", 0);
    const bool explicit_size = want > 0;
    if (want <= 0) want = std::max<long long>(6ll << 30, std::max(num_nvl_bytes, num_rdma_bytes));
    region_bytes = (size_t)((want - kCtrlBytes) / 6) & ~(size_t)4095;
    EP_HOST_ASSERT_S(region_bytes >= (1u << 20), "DEEPEP_WINDOW_BYTES too small: ", want);
    constexpr size_t kMaxRegionBytes = (size_t)1016 << 20;       // 2 regions per segment < 2040 MiB
    if (region_bytes > kMaxRegionBytes) {
        if (explicit_size && rank == 0)
            std::fprintf(stderr, "[deep_ep] DEEPEP_WINDOW_BYTES=%lld asks for %zu-byte regions; clamped to %zu (an ipc-mapped allocation "
                         "must stay below 2 GiB): use DEEPEP_NORMAL_LONG_SEQ_ROUND for larger batches\n", want, region_bytes, kMaxRegionBytes);
        region_bytes = kMaxRegionBytes;
    }
    seg_bytes[kSegCtrl] = (size_t)kCtrlBytes;
    for (int f = 0; f < 3; ++f) seg_bytes[1 + f] = 2 * region_bytes;
    window_bytes = kCtrlBytes + 6 * (int64_t)region_bytes;
    const bool want_fine = get_value_from_env("DEEPEP_WINDOW_FINEGRAINED", 1) != 0;
    // The protocol has running kernels poll flag / granule words that peer GPUs write and read rows peers wrote: that needs
    // fine-grained (system-coherent) memory.  A failed fine-grained allocation is an error, not a silent downgrade; a
    // coarse-grained window exists only on explicit request (DEEPEP_WINDOW_FINEGRAINED=0) and is reported through
    // is_window_fine_grained() so that deep_ep.Buffer keeps W > 1 traffic on the alltoall (RCCL) strategies.
    for (int sg = 0; sg < kNumSegs; ++sg) {
        void *p = nullptr;
        if (want_fine) {
            const hipError_t e = hipExtMallocWithFlags(&p, seg_bytes[sg], hipDeviceMallocFinegrained);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                throw EPException("HIP Assertion", __FILE__, __LINE__,
                                  ep_concat("cannot allocate a fine-grained window segment of ", seg_bytes[sg], " bytes (", hipGetErrorString(e),
                                            "); lower DEEPEP_WINDOW_BYTES, or set DEEPEP_WINDOW_FINEGRAINED=0 to run on the alltoall strategies"));
            }
        } else {
            HIP_CHECK(hipMalloc(&p, seg_bytes[sg]));
        }
        seg_base[sg] = (uint8_t *)p;
    }
    window_fine_grained = want_fine;
    window = seg_base[kSegCtrl];
    HIP_CHECK(hipMemset(window, 0, (size_t)kCtrlBytes));
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipHostMalloc((void **)&summary_host, sizeof(int32_t) * (4 + 2048), hipHostMallocMapped));
    HIP_CHECK(hipHostMalloc((void **)&status_host, sizeof(int32_t) * 4, hipHostMallocMapped));
    std::memset(summary_host, 0, sizeof(int32_t) * (4 + 2048));
    std::memset(status_host, 0, sizeof(int32_t) * 4);
    HIP_CHECK(hipHostGetDevicePointer((void **)&summary_dev, summary_host, 0));
    HIP_CHECK(hipHostGetDevicePointer((void **)&status_dev, status_host, 0));
    // how this device deals workgroups to its XCDs (8 on an MI355X: block b on XCD b % 8; 1 = nothing assumed): the requantising GEMM1 of
    // fused_deep_moe forms its workers from it (moe_gemm.hip)
    gemm_xcds = mi_ep_moe_probe_xcds(nullptr);
    peer_seg.assign((size_t)num_ranks, std::array<uint8_t *, kNumSegs>{});
    peer_opened.assign((size_t)num_ranks, false);
    for (int sg = 0; sg < kNumSegs; ++sg) peer_seg[(size_t)rank][(size_t)sg] = seg_base[sg];
    if (num_ranks == 1) available = true;      // nothing to exchange
    // normal-dispatch transport: remote writes ("push", default for W > 1: posted stores need no round trip over xGMI) or
    // remote reads ("pull"); both produce identical results, DEEPEP_DISPATCH_TRANSPORT / set_dispatch_transport() selects
    const char *tr = std::getenv("DEEPEP_DISPATCH_TRANSPORT");
    set_dispatch_transport(tr && *tr ? std::string(tr) : std::string(num_ranks > 1 ? "push" : "pull"));
}

// MI355X only: whether the rows of a rank's OWN tokens bypass the window (dispatch: gathered token by token by pull_local; combine:
// read in place by the reduce).  Switching both off makes every row take the path a REMOTE row takes -- on one GPU that is the
// kernel mix of an EP = 8 rank (7/8 of whose rows are remote), which bench.py times as `ep8_proxy`.  Results are identical either way.
void Buffer::set_local_row_paths(bool dispatch_local, bool combine_local)
{
    dispatch_local_rows = dispatch_local;
    combine_local_rows_enabled = combine_local;
}

void Buffer::set_dispatch_transport(const std::string &name)
{
    EP_HOST_ASSERT_S(name == "push" || name == "pull", "DEEPEP_DISPATCH_TRANSPORT must be push or pull, got ", name);
    dispatch_transport = name == "push" ? kTransportPush : kTransportPull;
}

Buffer::~Buffer() noexcept(false)
{
    hipDeviceSynchronize();
    for (size_t r = 0; r < peer_seg.size(); ++r)
        if (peer_opened[r])
            for (uint8_t *p : peer_seg[r])
                if (p) hipIpcCloseMemHandle(p);
    for (uint8_t *p : seg_base)
        if (p) hipFree(p);
    if (summary_host) hipHostFree(summary_host);
    if (status_host) hipHostFree(status_host);
}

std::string Buffer::get_local_ipc_handle() const
{
    std::string all;
    for (int sg = 0; sg < kNumSegs; ++sg) {
        hipIpcMemHandle_t h;
        HIP_CHECK(hipIpcGetMemHandle(&h, seg_base[sg]));
        all.append((const char *)&h, sizeof(h));
    }
    return all;
}

std::vector<int64_t> Buffer::get_local_window_ptrs() const
{
    std::vector<int64_t> v;
    for (uint8_t *p : seg_base) v.push_back((int64_t)p);
    return v;
}

void Buffer::sync(const std::vector<std::string> &handles, const std::vector<std::vector<int64_t>> &local_ptrs)
{
    EP_HOST_ASSERT((int64_t)handles.size() == num_ranks and (int64_t)local_ptrs.size() == num_ranks);
    for (int64_t r = 0; r < num_ranks; ++r) {
        if (r == rank) continue;
        if (!local_ptrs[(size_t)r].empty()) {                // the peer lives in this process: plain pointers
            EP_HOST_ASSERT((int)local_ptrs[(size_t)r].size() == kNumSegs);
            for (int sg = 0; sg < kNumSegs; ++sg) peer_seg[(size_t)r][(size_t)sg] = (uint8_t *)local_ptrs[(size_t)r][(size_t)sg];
            continue;
        }
        EP_HOST_ASSERT_S(handles[(size_t)r].size() == kNumSegs * sizeof(hipIpcMemHandle_t), "bad ipc handle from rank ", r);
        for (int sg = 0; sg < kNumSegs; ++sg) {
            hipIpcMemHandle_t h;
            std::memcpy(&h, handles[(size_t)r].data() + (size_t)sg * sizeof(h), sizeof(h));
            void *p = nullptr;
            HIP_CHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
            peer_seg[(size_t)r][(size_t)sg] = (uint8_t *)p;
        }
        peer_opened[(size_t)r] = true;
    }
    available = true;
}

// Two rounds of {flag, 4 KiB row, granule} with every peer through the mapped windows on the SAME addresses (write path, read-back
// path, 8-byte granule path; checked with ordinary cached loads, so a line cached in round 1 and served stale in round 2 shows
// up).  Called by deep_ep.Buffer on every rank right after sync(); a failure (no peer access, stale mapping, stores that never
// become visible) makes the Python side fall back to the alltoall (RCCL) strategies instead of corrupting data.
bool Buffer::self_test(int64_t test_timeout_ms)
{
    require_available();
    if (num_ranks == 1) return true;
    hipStream_t st = cur_stream();
    constexpr int kRounds = 2;
    const uint64_t ep = selftest_epoch + 1;
    selftest_epoch += kRounds;
    auto rows = peer_family_bases(kLLDispatch);        // scratch: the low-latency segment, unused before the first call
    auto flag_peers = peer_ptrs((size_t)(kOffFlags + kFlagSelfTest * kFlagGroupSlots * 8));
    auto ack_peers = peer_ptrs((size_t)(kOffFlags + kFlagSelfTestAck * kFlagGroupSlots * 8));
    EP_HOST_ASSERT(mi_ep_selftest_bytes((int)num_ranks) <= region_bytes);
    MI_EP_CHECK(mi_ep_selftest(rows.data(), (uint64_t *const *)flag_peers.data(),
                               (const uint64_t *)(window + kOffFlags + kFlagSelfTest * kFlagGroupSlots * 8),
                               (uint64_t *const *)ack_peers.data(),
                               (const uint64_t *)(window + kOffFlags + kFlagSelfTestAck * kFlagGroupSlots * 8), (int)num_ranks, (int)rank, ep,
                               kRounds, (uint32_t)(0x5E1F0000u + ep), status_dev, (int)test_timeout_ms, st));
    HIP_CHECK(hipStreamSynchronize(st));
    const int32_t code = __atomic_load_n(status_host, __ATOMIC_ACQUIRE);
    if (code != 0) {
        __atomic_store_n(status_host, 0, __ATOMIC_RELEASE);
        std::fprintf(stderr, "[deep_ep rank %lld] window self-test failed with code %d\n", (long long)rank, code);
        return false;
    }
    return true;
}

bool Buffer::self_test_in_launch(int64_t test_timeout_ms, int64_t skip_payload_from_round)
{
    require_available();
    if (num_ranks == 1) return true;
    hipStream_t st = cur_stream();
    constexpr int kRounds = 4;                           // both ping-pong halves twice: every address is rewritten under a reader's caches
    const uint64_t ep = selftest_epoch + 1;
    selftest_epoch += kRounds;
    const int W = (int)num_ranks;
    const size_t row_bytes = mi_ep_selftest_inlaunch_bytes(W), flag_words = mi_ep_selftest_inlaunch_flag_words(W);
    EP_HOST_ASSERT(row_bytes <= region_bytes && flag_words * 4 <= (size_t)kRowFlagsParityBytes);
    // scratch: the head of both halves of the low-latency segment (unused before the first call) and the LAST words of both halves of the
    // row-flag area (the combine indexes it by t * K + k from 0); both are cleared again below
    auto rows = peer_family_bases(kLLDispatch);
    const size_t flag_off = (size_t)kRowFlagsParityBytes - flag_words * 4;
    auto flag_peers = peer_ptrs((size_t)kOffRowFlags + flag_off);
    auto ack_peers = peer_ptrs((size_t)(kOffFlags + kFlagSelfTestAck * kFlagGroupSlots * 8));
    MI_EP_CHECK(mi_ep_selftest_inlaunch(rows.data(), region_bytes, (uint32_t *const *)flag_peers.data(), (size_t)kRowFlagsParityBytes,
                                        (uint64_t *const *)ack_peers.data(),
                                        (const uint64_t *)(window + kOffFlags + kFlagSelfTestAck * kFlagGroupSlots * 8), W, (int)rank, ep, kRounds,
                                        (uint32_t)(0x1A7C0000u + ep), (int)skip_payload_from_round, status_dev, (int)test_timeout_ms, st));
    HIP_CHECK(hipStreamSynchronize(st));
    // (this rank's consumers have checked -- or given up on -- every row and word its peers write: its scratch can be cleared.  The tags and
    //  flag words left there must not meet a call of the product whose epoch happens to match.)
    for (int h = 0; h < 2; ++h) {
        HIP_CHECK(hipMemsetAsync(family_base(kLLDispatch) + (size_t)h * region_bytes, 0, row_bytes, st));
        HIP_CHECK(hipMemsetAsync(window + kOffRowFlags + (size_t)h * kRowFlagsParityBytes + flag_off, 0, flag_words * 4, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    const int32_t code = __atomic_load_n(status_host, __ATOMIC_ACQUIRE);
    if (code != 0) {
        __atomic_store_n(status_host, 0, __ATOMIC_RELEASE);
        std::fprintf(stderr, "[deep_ep rank %lld] in-launch hand-off self-test failed with code %d\n", (long long)rank, code);
        return false;
    }
    return true;
}

std::vector<std::pair<double, double>> Buffer::get_gemm_clock() const
{
    HIP_CHECK(hipDeviceSynchronize());
    double ghz[3], us[3];
    MI_EP_CHECK(mi_ep_moe_gemm_clock(ghz, us));
    return {{ghz[0], us[0]}, {ghz[1], us[1]}, {ghz[2], us[2]}};
}

void Buffer::require_available() const
{
    if (!available)
        throw EPException("Assertion", __FILE__, __LINE__,
                          "deep_ep_cpp.Buffer used before sync(): peers' windows are not mapped");
}

// base (ping-pong half 0) of this rank's segment of `family`; the kernels add (epoch & 1) * region_bytes themselves
uint8_t *Buffer::family_base(int family) const { return seg_base[(size_t)(1 + family)]; }

// device-resident count of completed calls of `family` (own control segment; see include/mi_ep.h "Device-resident epochs")
uint64_t *Buffer::epoch_ctr(int family) const { return (uint64_t *)(window + kOffEpochs) + family; }

// every rank's pointer to byte `offset` of the control segment
std::vector<void *> Buffer::peer_ptrs(size_t offset) const
{
    std::vector<void *> v((size_t)num_ranks);
    for (size_t r = 0; r < (size_t)num_ranks; ++r) v[r] = peer_seg[r][kSegCtrl] + offset;
    return v;
}

// every rank's segment base of `family`
std::vector<void *> Buffer::peer_family_bases(int family) const
{
    std::vector<void *> v((size_t)num_ranks);
    for (size_t r = 0; r < (size_t)num_ranks; ++r) v[r] = peer_seg[r][(size_t)(1 + family)];
    return v;
}

void Buffer::check_status(const char *where)
{
    const int32_t s = __atomic_load_n(status_host, __ATOMIC_ACQUIRE);
    if (s != 0) {
        __atomic_store_n(status_host, 0, __ATOMIC_RELEASE);
        if (s == MI_EP_STATUS_GEMM_ROWMAX) fused_requant = false;      // the column tiles of a row block never met: two launches from now on
        if (s == MI_EP_STATUS_LAYOUT_BARRIER) {
            // a barrier that did not close never re-armed its pair of sync words: clear the whole ring (behind everything queued on this
            // stream) so the pair does not hand a non-zero count to the launch that borrows it 32 layout calls later
            (void)hipMemsetAsync(window + kOffEpochs + 512, 0, 32 * 8, cur_stream());
        }
        throw EPException("Timeout", __FILE__, __LINE__,
                          ep_concat(where, ": a peer did not arrive within DEEPEP_TIMEOUT_MS=", timeout_ms, " (code ", s,
                                    ", rank ", rank, ")"));
    }
}

int64_t Buffer::wait_summary(const char *where)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
        const int32_t v = __atomic_load_n(summary_host, __ATOMIC_ACQUIRE);
        if (v >= 0) return v;
        if ((spins & 0xFFF) == 0xFFF) {
            check_status(where);
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 2ll * timeout_ms)
                throw EPException("Timeout", __FILE__, __LINE__, ep_concat(where, ": notify summary never arrived"));
            hipError_t q = hipStreamQuery(cur_stream());
            if (q != hipSuccess && q != hipErrorNotReady) HIP_CHECK(q);
        }
    }
}

// one word of the pinned summary: normally there already when the total (word 0) is, but the kernel orders nothing
int32_t Buffer::summary_word(int i, const char *where)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; ++spins) {
        const int32_t v = __atomic_load_n(summary_host + i, __ATOMIC_ACQUIRE);
        if (v >= 0) return v;
        if ((spins & 0xFFF) == 0xFFF) {
            check_status(where);
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > 2ll * timeout_ms) throw EPException("Timeout", __FILE__, __LINE__, ep_concat(where, ": notify summary word ", i, " never arrived"));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A1  get_dispatch_layout  (reference deep_ep.cpp:111-180)
// ------------------------------------------------------------------------------------------------
Buffer::Layout Buffer::run_layout(const at::Tensor &topk_idx, int num_experts)
{
    EP_HOST_ASSERT(topk_idx.dim() == 2);
    EP_HOST_ASSERT(topk_idx.is_contiguous());
    EP_HOST_ASSERT(topk_idx.is_cuda());
    EP_HOST_ASSERT(topk_idx.scalar_type() == at::kLong or topk_idx.scalar_type() == at::kInt);
    EP_HOST_ASSERT(num_experts > 0);
    EP_HOST_ASSERT_S(num_experts % num_ranks == 0, "num_experts (", num_experts, ") must be a multiple of num_ranks (",
                     num_ranks, ")");
    const int T = (int)topk_idx.size(0), K = (int)topk_idx.size(1);
    EP_HOST_ASSERT_S(T >= 0 && T <= kMaxTotalTokens, "num_tokens (", T, ") must be in the range [0, ", kMaxTotalTokens, "].");
    EP_HOST_ASSERT_S(K >= 1 && K <= MI_EP_MAX_TOPK, "num_topk (", K, ") must be in [1, ", MI_EP_MAX_TOPK, "]");
    auto i32 = at::dtype(at::kInt).device(topk_idx.device());
    Layout l;
    l.T = T, l.K = K, l.E = num_experts, l.idx = topk_idx, l.idx_version = Layout::version_of(topk_idx);
    l.num_tokens_per_expert = at::empty({num_experts}, i32);
    l.num_tokens_per_rank = at::empty({num_ranks}, i32);
    l.is_token_in_rank = at::empty({T, num_ranks}, i32);
    l.send_token_idx_small = at::empty({T, K}, i32);
    l.send_data_offset = at::empty({num_experts}, i32);
    const size_t wsb = mi_ep_dispatch_layout_workspace(T, K, num_experts);
    l.workspace = at::empty({(int64_t)wsb}, at::dtype(at::kByte).device(topk_idx.device()));
    MI_EP_CHECK(mi_ep_dispatch_layout(topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, T, K, num_experts,
                                      (int)num_ranks, l.num_tokens_per_rank.data_ptr<int>(),
                                      l.num_tokens_per_expert.data_ptr<int>(), l.is_token_in_rank.data_ptr<int>(),
                                      l.send_token_idx_small.data_ptr<int>(), l.send_data_offset.data_ptr<int>(),
                                      l.workspace.data_ptr(), wsb, layout_sync_words(topk_idx.device()), status_dev, cur_stream()));
    return l;
}

// two persistent words for the cooperative layout launch: in the rank's control area (zeroed at construction, behind the per-family
// call counters), so no allocation can happen under a graph capture; the kernel's grid barrier re-arms them itself.  A pair serves
// ONE launch in flight: calls that overlap on different streams of one Buffer (two-batch overlap) must not share it, so the pairs are
// dealt from a ring of kLayoutSyncPairs -- a pair comes round again only after that many later layout calls were issued.  (Under a
// graph capture the pair is baked into the graph: replays of ONE graph serialise on its stream, and eager calls keep moving through
// the ring.)  A barrier that does not close within 2 s reports MI_EP_STATUS_LAYOUT_BARRIER: check_status raises.
uint32_t *Buffer::layout_sync_words(const at::Device &)
{
    constexpr uint64_t kLayoutSyncPairs = 32;         // 32 x 8 B at control offset kOffEpochs + 512 .. + 768
    const uint64_t i = layout_calls++ % kLayoutSyncPairs;
    return (uint32_t *)(window + kOffEpochs + 512) + 2 * i;
}

const Buffer::Layout &Buffer::layout_for(const at::Tensor &topk_idx, int num_experts)
{
    // The reference silently reuses whatever get_dispatch_layout stashed last (deep_ep.cpp:170-172,321).  We reuse the
    // stash only when it was computed for this very tensor; otherwise the layout is recomputed (one extra ~30 us
    // kernel chain), which removes the hidden ordering requirement without changing results.
    // "This very tensor" = same storage address AND the stash still holds that storage (so the address cannot have been
    // recycled for another tensor by the caching allocator) AND the same version counter (no in-place rewrite since).
    if (!stash.matches(topk_idx, num_experts)) stash = run_layout(topk_idx, num_experts);
    return stash;
}

std::tuple<at::Tensor, std::optional<at::Tensor>, at::Tensor, at::Tensor, std::optional<EventHandle>>
Buffer::get_dispatch_layout(const at::Tensor &topk_idx, int num_experts, std::optional<EventHandle> &, bool, bool)
{
    stash = run_layout(topk_idx, num_experts);
    return {stash.num_tokens_per_rank, std::nullopt, stash.num_tokens_per_expert, stash.is_token_in_rank, std::nullopt};
}

at::Tensor Buffer::get_notify_send_data()
{
    // The reference returns an Ascend910B-only staging payload (deep_ep.cpp:142-164,182-185).  The part of it that is
    // meaningful on a single xGMI node -- "the number of tokens every expert receives from this rank" -- is returned.
    EP_HOST_ASSERT_S(stash.T >= 0, "get_dispatch_layout has not been called");
    return stash.num_tokens_per_expert;
}

void Buffer::clean_low_latency_buffer(int, int, int) {}   // no-op, as in the reference (buffer.py:267-283)

// ------------------------------------------------------------------------------------------------
// A2 + A3  intranode_dispatch  (reference deep_ep.cpp:197-416)
// ------------------------------------------------------------------------------------------------
// Device half of a normal-mode dispatch, shared by intranode_dispatch and the prefill-size leg of fused_deep_moe:
// stage (push: into the destination ranks' windows; pull: into the own window) -> ONE single-workgroup launch that posts
// this rank's counts + "rows staged" flag, collects everybody's and derives the tables.  The call's epoch and ping-pong
// half come from the device-resident counter of the family (see include/mi_ep.h), so nothing here depends on how many
// calls ran before.
Buffer::DispatchExchange Buffer::dispatch_exchange(const at::Tensor &x, const at::Tensor &topk_idx, const Layout &lay, int E, int qm,
                                                   bool want_summary, int32_t *wait_stats, hipStream_t st)
{
    const int T = (int)x.size(0), H = (int)x.size(1), K = (int)topk_idx.size(1), W = (int)num_ranks, L = E / W;
    const size_t rb = mi_ep_dispatch_row_bytes(H, qm);
    DispatchExchange ex;
    ex.push = dispatch_transport == kTransportPush;
    ex.slab_bytes = ex.push ? mi_ep_dispatch_push_slab_bytes(region_bytes, W) : region_bytes;
    // compact staging: one row per TOKEN plus K index entries, not one row per (t, k).  Transport "pull": rows staged in the own
    // window, receivers read them over xGMI (mi_ep_dispatch_stage_compact); "push": rows written once into every destination
    // rank's window, slab `rank` of its region (mi_ep_dispatch_stage_push), receivers gather locally.
    EP_HOST_ASSERT_S((size_t)T <= mi_ep_dispatch_index_offset(H, qm, K, ex.slab_bytes) / rb, "dispatch window too small: need ",
                     (size_t)T * (rb + (size_t)K * 8) * (ex.push ? (size_t)W : 1), " bytes per region, have ", region_bytes,
                     "; raise DEEPEP_WINDOW_BYTES (or use DEEPEP_NORMAL_LONG_SEQ_ROUND)");
    EP_HOST_ASSERT_S(mi_ep_notify_lds_bytes(W, E) <= 160u * 1024u, "num_experts (", E, ") x num_ranks (", W,
                     "): the count exchange keeps W * (E + 1) counts in one workgroup's LDS (", mi_ep_notify_lds_bytes(W, E),
                     " bytes needed, 163840 available); use fewer experts per exchange");
    uint64_t *ctr = epoch_ctr(kDispatch);
    auto region_peers = peer_family_bases(kDispatch);
    const bool i32idx = topk_idx.scalar_type() == at::kInt;
    { ProfScope ps_(this, ex.push ? "dispatch_stage_push" : "dispatch_stage", st);
      if (ex.push)
          MI_EP_CHECK(mi_ep_dispatch_stage_push(x.data_ptr(), topk_idx.data_ptr(), i32idx, lay.send_token_idx_small.data_ptr<int>(),
                                                lay.send_data_offset.data_ptr<int>(), T, K, H, E, W, (int)rank, qm,
                                                region_peers.data(), region_bytes, ctr, region_bytes, st));
      else
          MI_EP_CHECK(mi_ep_dispatch_stage_compact(x.data_ptr(), topk_idx.data_ptr(), i32idx, lay.send_token_idx_small.data_ptr<int>(),
                                                   lay.send_data_offset.data_ptr<int>(), T, K, H, E, (int)rank, qm,
                                                   family_base(kDispatch), region_bytes, ctr, region_bytes, st)); }
    auto notify_peers = peer_ptrs((size_t)kOffNotify);
    auto flag_peers = peer_ptrs((size_t)(kOffFlags + kFlagDispatch * kFlagGroupSlots * 8));
    ex.nt = alloc_notify_tables(W, E, L, at::dtype(at::kInt).device(x.device()));
    NotifyTables &nt = ex.nt;
    if (want_summary) {        // every word the host will read: the kernel writes them in no particular order (mi_ep.h)
        for (int i = 2 + L - 1; i >= 1; --i) __atomic_store_n(summary_host + i, -1, __ATOMIC_RELAXED);
        __atomic_store_n(summary_host, -1, __ATOMIC_RELEASE);
    }
    { ProfScope ps_(this, "dispatch_notify", st);
      MI_EP_CHECK(mi_ep_notify_exchange_tables((uint64_t *const *)notify_peers.data(), (uint64_t *const *)flag_peers.data(),
                                               lay.num_tokens_per_expert.data_ptr<int>(), T, (const uint64_t *)(window + kOffNotify), 0,
                                               (const uint64_t *)(window + kOffFlags + kFlagDispatch * kFlagGroupSlots * 8), 0,
                                               nt.cnt.data_ptr<int>(), W, E, (int)rank, ex.push ? 1 : 0, nt.recv_count.data_ptr<int>(),
                                               nt.recv_offset.data_ptr<int>(), nt.recv_tokens_per_expert.data_ptr<int>(),
                                               nt.expert_global_offset.data_ptr<int>(), nt.srcrank_in_expert_offset.data_ptr<int>(),
                                               nt.r_in_srcrank_offset.data_ptr<int>(), nt.total_recv_token.data_ptr<int>(),
                                               nt.max_bs.data_ptr<int>(), nt.pull_offset.data_ptr<int>(),
                                               want_summary ? summary_dev : nullptr, ctr, (size_t)kNotifyParityBytes, status_dev,
                                               timeout_ms, wait_stats, st)); }
    // pull: token rows + index live in every SOURCE rank's window; push: in the source slabs of the own window
    ex.src_bases = region_peers;
    if (ex.push)
        for (int s = 0; s < W; ++s) ex.src_bases[(size_t)s] = family_base(kDispatch) + (size_t)s * ex.slab_bytes;
    ex.topk_idx = topk_idx, ex.send_token_idx_small = lay.send_token_idx_small, ex.num_tokens_per_expert = lay.num_tokens_per_expert;
    ex.num_tokens = T, ex.num_experts = E;
    return ex;
}

// receiver half: gather `rows_alloc` rows (at most; the device-side total bounds it too) into freshly allocated outputs
void Buffer::dispatch_pull(const DispatchExchange &ex, int H, int K, int L, int qm, int64_t rows_alloc, const at::TensorOptions &x_opts,
                           at::Tensor &rx, at::Tensor &rs, at::Tensor &src_idx, hipStream_t st)
{
    auto dev = x_opts.device();
    const bool quant = qm != MI_EP_QUANT_NONE;
    rx = quant ? at::empty({rows_alloc, H}, at::dtype(payload_dtype(qm)).device(dev)) : at::empty({rows_alloc, H}, x_opts);
    rs = at::empty({rows_alloc}, at::dtype(at::kFloat).device(dev));
    src_idx = at::empty({rows_alloc * 3}, at::dtype(at::kInt).device(dev));
    ProfScope ps_(this, "dispatch_pull", st);
    // this rank's own tokens are gathered token by token (their staged row is read once, not once per selection); everybody else's
    // rows row by row.  MI_EP_DISPATCH_LOCAL=0: every row through pull_indexed.
    const bool local = dispatch_local_rows;
    // At EP = 1 every received row is one of this rank's own: pull_local also records, per (token, selection), the receive row it wrote.
    // That is the table the combine push would build later, so the combine of this exchange needs no push and no signal / wait at all
    // (remember_local_rows / intranode_combine).
    at::Tensor local_row_out;
    if (local && num_ranks == 1 && combine_local_rows_enabled && ex.num_tokens > 0)
        local_row_out = at::empty({(int64_t)ex.num_tokens * K}, at::dtype(at::kInt).device(dev));
    if (local && ex.num_tokens > 0)
        MI_EP_CHECK(mi_ep_dispatch_pull_local(ex.src_bases[(size_t)rank], ex.topk_idx.data_ptr(), ex.topk_idx.scalar_type() == at::kInt,
                                              ex.send_token_idx_small.data_ptr<int>(), ex.nt.recv_count.data_ptr<int>(),
                                              ex.num_tokens_per_expert.data_ptr<int>(), ex.num_tokens, K, H, ex.num_experts, (int)num_ranks,
                                              (int)rank, qm, (int)rows_alloc, rx.data_ptr(), quant ? rs.data_ptr<float>() : nullptr,
                                              src_idx.data_ptr<int>(), local_row_out.defined() ? local_row_out.data_ptr<int>() : nullptr,
                                              epoch_ctr(kDispatch), region_bytes, st));
    if (local_row_out.defined()) remember_local_rows(src_idx, local_row_out, ex.num_tokens, K);
    if (!(local && num_ranks == 1))
        MI_EP_CHECK(mi_ep_dispatch_pull_indexed((const void *const *)ex.src_bases.data(), ex.nt.recv_count.data_ptr<int>(),
                                                ex.nt.pull_offset.data_ptr<int>(), (int)num_ranks, L, H, K, qm, (int)rows_alloc,
                                                ex.slab_bytes, rx.data_ptr(), quant ? rs.data_ptr<float>() : nullptr,
                                                src_idx.data_ptr<int>(), epoch_ctr(kDispatch), region_bytes, local ? (int)rank : -1, st));
}

std::tuple<at::Tensor, std::optional<at::Tensor>, std::optional<at::Tensor>, std::optional<at::Tensor>, std::vector<int>,
           at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, std::optional<EventHandle>>
Buffer::intranode_dispatch(const at::Tensor &x, const std::optional<at::Tensor> &x_scales,
                           const std::optional<at::Tensor> &topk_idx, const std::optional<at::Tensor> &topk_weights,
                           const std::optional<at::Tensor> &num_tokens_per_rank, const at::Tensor &is_token_in_rank,
                           const std::optional<at::Tensor> &num_tokens_per_expert, int, const std::optional<at::Tensor> &,
                           const std::optional<at::Tensor> &, const std::optional<at::Tensor> &dispatch_wait_recv_cost_stats,
                           int expert_alignment, int num_worst_tokens, const Config &config,
                           std::optional<EventHandle> &, bool, bool, bool use_quant, const std::string &quant_type)
{
    require_available();
    EP_HOST_ASSERT(config.num_sms % 2 == 0);
    const int num_channels = config.num_sms / 2;
    EP_HOST_ASSERT(num_tokens_per_rank.has_value());
    EP_HOST_ASSERT(num_tokens_per_expert.has_value());
    EP_HOST_ASSERT(num_tokens_per_expert->scalar_type() == at::kInt);
    EP_HOST_ASSERT(num_tokens_per_rank->scalar_type() == at::kInt);
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous());
    EP_HOST_ASSERT(x.scalar_type() == at::kBFloat16);
    EP_HOST_ASSERT(!x_scales.has_value());      // pre-quantised MXFP8 input is Ascend950-only
    EP_HOST_ASSERT(num_tokens_per_expert->dim() == 1 and num_tokens_per_expert->is_contiguous());
    EP_HOST_ASSERT(num_tokens_per_expert->size(0) % num_ranks == 0);
    EP_HOST_ASSERT(num_tokens_per_rank->dim() == 1 and num_tokens_per_rank->is_contiguous());
    EP_HOST_ASSERT(num_tokens_per_rank->size(0) == num_ranks);
    EP_HOST_ASSERT(expert_alignment == 1);
    (void)is_token_in_rank;
    const int T = (int)x.size(0), H = (int)x.size(1);
    const int E = (int)num_tokens_per_expert->size(0);
    const int W = (int)num_ranks, L = E / W;
    EP_HOST_ASSERT(topk_idx.has_value());
    EP_HOST_ASSERT(topk_weights.has_value());
    EP_HOST_ASSERT(topk_idx->dim() == 2 and topk_idx->is_contiguous());
    EP_HOST_ASSERT(topk_weights->dim() == 2 and topk_weights->is_contiguous());
    EP_HOST_ASSERT(T == topk_idx->size(0));
    const int K = (int)topk_idx->size(1);
    EP_HOST_ASSERT(K == topk_weights->size(1));
    EP_HOST_ASSERT(topk_weights->scalar_type() == at::kFloat);
    EP_HOST_ASSERT_S(H % 16 == 0 && H <= MI_EP_MAX_HIDDEN, "hidden (", H, ") must be a multiple of 16 and <= ", MI_EP_MAX_HIDDEN);
    EP_HOST_ASSERT_S(E <= 2048, "num_experts (", E, ") must be <= 2048");
    const int qm = quant_mode_of(use_quant, quant_type);
    check_status("intranode_dispatch");
    ++profile_calls;
    const Layout &lay = layout_for(*topk_idx, E);
    hipStream_t st = cur_stream();
    auto i32 = at::dtype(at::kInt).device(x.device());

    // counts + "staged" flag to every peer, then (same launch, one workgroup) everybody's counts + flags -> tables
    // (+ pinned summary for the host)
    int32_t *wait_stats = nullptr;
    if (dispatch_wait_recv_cost_stats.has_value()) {
        EP_HOST_ASSERT(dispatch_wait_recv_cost_stats->scalar_type() == at::kInt and dispatch_wait_recv_cost_stats->is_contiguous());
        EP_HOST_ASSERT(dispatch_wait_recv_cost_stats->dim() == 1 and dispatch_wait_recv_cost_stats->size(0) == num_ranks);
        wait_stats = dispatch_wait_recv_cost_stats->data_ptr<int>();
    }
    const bool host_sync = num_worst_tokens <= 0;
    DispatchExchange ex = dispatch_exchange(x, *topk_idx, lay, E, qm, host_sync, wait_stats, st);

    // Receive buffers + the pull launch.  The exact row count reaches the host only through the pinned summary word;
    // to keep the GPU busy across that round trip the pull is launched FIRST into buffers sized from the previous call
    // (+25 %), and the results are returned as exact-size prefixes once the host knows the count.  A call that receives
    // more than the guess simply pulls again into exact-size buffers (the kernel never writes past `rows_hint`).
    at::Tensor expandx_out, dynamic_scales_out, expand_idx_out;
    static const bool speculate = get_value_from_env("DEEPEP_SPECULATIVE_RECV", 1) != 0;
    int64_t guess = 0;
    if (host_sync && speculate && last_recv_rows > 0) {
        guess = last_recv_rows + last_recv_rows / 4 + 256;
        dispatch_pull(ex, H, K, L, qm, guess, x.options(), expandx_out, dynamic_scales_out, expand_idx_out, st);
    }
    // Everything the host can prepare without the row count is prepared BEFORE it waits for it: the wait ends when the notify kernel
    // does, and from then on the host has one pull kernel (~40 us at 4096 tokens) to return, get called again and queue the combine.
    // (Measured: 3-4 us from the end of the wait to the return, 9-11 us of Python until the combine is entered, 4-5 us until its kernel
    // is queued -- 18 us of the 40; a 20 us busy-wait injected after the wait did not change the step time.  The 6-8 us of idle GPU in
    // front of the reduce that a rocprofv3 kernel trace shows is the tracer's own per-call cost.)
    // placeholders kept for handle-shape compatibility (uninitialised in the reference, deep_ep.cpp:220-222)
    auto rank_prefix_matrix = at::empty({W, W}, i32);
    auto channel_prefix_matrix = at::empty({W, num_channels}, i32);
    auto recv_channel_prefix_matrix = at::empty({W, num_channels}, i32);
    at::Tensor recv_idx_guess, recv_w_guess;           // allocated, never written (deep_ep.cpp:371-374)
    if (guess > 0) {
        recv_idx_guess = at::empty({guess, K}, topk_idx->options());
        recv_w_guess = at::empty({guess, K}, topk_weights->options());
    }
    int64_t trt;
    std::vector<int> num_recv_tokens_per_expert_list;
    num_recv_tokens_per_expert_list.reserve((size_t)L);
    const int token_nums_type = get_value_from_env("MOE_EXPERT_TOKEN_NUMS_TYPE", 1);     // (read per call: callers switch it)
    if (host_sync) {
        trt = wait_summary("intranode_dispatch");
        check_status("intranode_dispatch");     // a peer that timed out inside THIS call's notify surfaces now, not one call later
        real_max_bs = summary_word(1, "intranode_dispatch");
        // counts, or inclusive cumsum when MOE_EXPERT_TOKEN_NUMS_TYPE=0 (deep_ep.cpp:311-312,384-401)
        const int type = token_nums_type;
        EP_HOST_ASSERT(type == 1 or type == 0);
        int run = 0;
        for (int le = 0; le < L; ++le) {
            const int c = summary_word(2 + le, "intranode_dispatch");
            run = (type == 0) ? run + c : c;
            num_recv_tokens_per_expert_list.push_back(run);
        }
        last_recv_rows = trt;
    } else {
        // DeepEP's graph-friendly mode: worst-case sized outputs, no host sync, empty list (buffer.py:337-338,356-358)
        trt = num_worst_tokens;
        // the largest per-rank batch is not read back in this mode; the caller's bound on the received ROWS (at most max_T * K rows
        // from each of W ranks) bounds it: max_T <= num_worst_tokens / (K * W), rounded up.  (Taking the row bound itself for a token
        // count made combine ask for K * W times the slot area it needs: C2 at one rank wanted 3.7 GB.)
        real_max_bs = std::max<int64_t>(real_max_bs, ((int64_t)num_worst_tokens + (int64_t)K * W - 1) / ((int64_t)K * W));
    }
    const int64_t rows = trt == 0 ? 1 : trt;      // deep_ep.cpp:327-328
    if (guess >= rows) {
        expandx_out = expandx_out.narrow(0, 0, rows);
        dynamic_scales_out = dynamic_scales_out.narrow(0, 0, rows);
        expand_idx_out = expand_idx_out.narrow(0, 0, rows * 3);
    } else {
        dispatch_pull(ex, H, K, L, qm, rows, x.options(), expandx_out, dynamic_scales_out, expand_idx_out, st);
    }
    std::optional<at::Tensor> recv_topk_idx, recv_topk_weights;                                // allocated, never written
    if (guess >= trt && guess > 0) {
        recv_topk_idx = recv_idx_guess.narrow(0, 0, trt);
        recv_topk_weights = recv_w_guess.narrow(0, 0, trt);
    } else {
        recv_topk_idx = at::empty({trt, K}, topk_idx->options());
        recv_topk_weights = at::empty({trt, K}, topk_weights->options());
    }
    return {expandx_out, dynamic_scales_out, recv_topk_idx, recv_topk_weights, num_recv_tokens_per_expert_list,
            rank_prefix_matrix, channel_prefix_matrix, recv_channel_prefix_matrix, expand_idx_out, ex.nt.recv_count, std::nullopt};
}

std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>
Buffer::notify_verify(const at::Tensor &x, const std::optional<at::Tensor> &, const std::optional<at::Tensor> &topk_idx,
                      const std::optional<at::Tensor> &, const std::optional<at::Tensor> &, const at::Tensor &,
                      const std::optional<at::Tensor> &num_tokens_per_expert, int, const std::optional<at::Tensor> &,
                      const std::optional<at::Tensor> &, const std::optional<at::Tensor> &, int, int, const Config &,
                      std::optional<EventHandle> &, bool, bool, bool use_quant)
{
    // Test-only entry of the reference (deep_ep.cpp:418-550): run the notify exchange and return its tables
    // (recv_data, recv_count, recv_offset, expert_global_offset, srcrank_in_expert_offset, r_in_srcrank_offset,
    //  total_recv_token, max_bs, recv_tokens_per_expert).  Here it is a complete dispatch exchange without the receive side (the
    // staged rows are simply never gathered), so the call counters of all ranks stay in step.
    require_available();
    EP_HOST_ASSERT(topk_idx.has_value() and num_tokens_per_expert.has_value());
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous() and x.scalar_type() == at::kBFloat16);
    const int E = (int)num_tokens_per_expert->size(0);
    const Layout &lay = layout_for(*topk_idx, E);
    DispatchExchange ex = dispatch_exchange(x, *topk_idx, lay, E, use_quant ? MI_EP_QUANT_INT8 : MI_EP_QUANT_NONE, false, nullptr,
                                            cur_stream());
    NotifyTables &nt = ex.nt;
    return {nt.cnt, nt.recv_count, nt.recv_offset, nt.expert_global_offset, nt.srcrank_in_expert_offset, nt.r_in_srcrank_offset,
            nt.total_recv_token, nt.max_bs, nt.recv_tokens_per_expert};
}

// ------------------------------------------------------------------------------------------------
// A4  intranode_combine  (reference deep_ep.cpp:552-608)
// ------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, std::optional<at::Tensor>, std::optional<EventHandle>>
Buffer::intranode_combine(const at::Tensor &x, const at::Tensor &topk_idx, const std::optional<at::Tensor> &topk_weights,
                          const at::Tensor &src_idx, const at::Tensor &send_head,
                          const std::optional<at::Tensor> &combine_send_cost_stats)
{
    require_available();
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous());
    EP_HOST_ASSERT(x.scalar_type() == at::kBFloat16);
    EP_HOST_ASSERT(topk_idx.dim() == 2 and topk_idx.is_contiguous());
    EP_HOST_ASSERT(src_idx.scalar_type() == at::kInt and send_head.scalar_type() == at::kInt);
    const int T = (int)topk_idx.size(0), K = (int)topk_idx.size(1), H = (int)x.size(1);
    const int W = (int)num_ranks, E = (int)send_head.size(0);
    if (topk_weights.has_value()) {
        EP_HOST_ASSERT(topk_weights->scalar_type() == at::kFloat and topk_weights->is_contiguous());
        EP_HOST_ASSERT(topk_weights->size(0) == T and topk_weights->size(1) == K);
    }
    if (combine_send_cost_stats.has_value()) {
        EP_HOST_ASSERT(combine_send_cost_stats->scalar_type() == at::kInt);
        EP_HOST_ASSERT(combine_send_cost_stats->dim() == 1 and combine_send_cost_stats->size(0) == num_ranks);
    }
    const size_t cb = mi_ep_combine_row_bytes(H);
    // every rank's slot area must fit the largest per-rank batch (real_max_bs from the matching dispatch)
    const int64_t max_rows = std::max<int64_t>((int64_t)T, real_max_bs) * K;
    EP_HOST_ASSERT_S((size_t)max_rows * cb <= region_bytes, "combine window too small: need ", (size_t)max_rows * cb,
                     " bytes per region, have ", region_bytes, "; raise DEEPEP_WINDOW_BYTES");
    check_status("intranode_combine");
    hipStream_t st = cur_stream();
    auto dst_peers = peer_family_bases(kCombine);
    // diagnose (opt-in): every send of this rank is complete when the push kernel ends, so each destination is charged the
    // push duration (device timestamps before / after; only launched when the caller passes the stats tensor)
    at::Tensor t_start;
    if (combine_send_cost_stats.has_value()) {
        EP_HOST_ASSERT(combine_send_cost_stats->is_contiguous());
        t_start = at::empty({1}, at::dtype(at::kLong).device(x.device()));
        MI_EP_CHECK(mi_ep_timestamp((uint64_t *)t_start.data_ptr(), st));
    }
    // EP = 1 and the handle comes from a dispatch that recorded its receive rows: the weighted sum reads x in place, nothing to push, nobody
    // to wait for (same values in the same order as the three-launch form: tests/ep_harness.py runs both against the oracle)
    if (W == 1 && combine_local_rows_enabled && !combine_send_cost_stats.has_value() && T > 0) {
        const at::Tensor known = recall_local_rows(src_idx, T, K);
        if (known.defined()) {
            at::Tensor out = at::empty({T, H}, x.options());
            ProfScope ps_(this, "combine_reduce", st);
            MI_EP_CHECK(mi_ep_combine_reduce(family_base(kCombine), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt,
                                             topk_weights.has_value() ? topk_weights->data_ptr<float>() : nullptr, nullptr, nullptr, T, K, H, E,
                                             out.data_ptr(), nullptr, 0, x.data_ptr(), known.data_ptr<int>(), (int)x.size(0), 0, 1, st));
            return {out, std::nullopt, std::nullopt};
        }
    }
    // total rows = send_head[E-1] (cam_moe_combine_normal.h:225), read on device
    // rows whose token lives on this rank stay where they are: the push only records their row number, the reduce reads x
    auto local_row = combine_local_rows(topk_idx);
    int signalled = 0;
    // (with send-cost statistics asked for, the push stays a launch of its own: the statistic is the time until the rows are out)
    if (combine_send_cost_stats.has_value()) {
        ProfScope ps_(this, "combine_push", st);
        MI_EP_CHECK(mi_ep_combine_push(x.data_ptr(), src_idx.data_ptr<int>(), send_head.data_ptr<int>() + (E - 1), (int)x.size(0), H, K,
                                       dst_peers.data(), W, region_bytes, epoch_ctr(kCombine), region_bytes, (int)rank,
                                       local_row.defined() ? local_row.data_ptr<int>() : nullptr, st));
        MI_EP_CHECK(mi_ep_elapsed_add(combine_send_cost_stats->data_ptr<int>(), W, (const uint64_t *)t_start.data_ptr(), st));
    } else {
        combine_push_rows(x, src_idx.data_ptr<int>(), send_head.data_ptr<int>() + (E - 1), (int)x.size(0), H, K, local_row, "combine_push", st,
                          signalled);
    }
    return {combine_finish(topk_idx, topk_weights.has_value() ? topk_weights->data_ptr<float>() : nullptr, H, E, x.options(),
                           "combine_reduce", st, x, local_row, signalled),
            std::nullopt, std::nullopt};
}

// Shared-expert ranks: the routing table the kernels see (include/mi_ep.h, mi_ep_shared_expert_map).  S == 0: the caller's own table.
Buffer::SharedView Buffer::shared_view(const at::Tensor &topk_idx, const float *weights, bool want_weights, int64_t num_experts,
                                       hipStream_t st) const
{
    const int W = (int)num_ranks, S = (int)shared_expert_rank_num, T = (int)topk_idx.size(0), K = (int)topk_idx.size(1);
    SharedView v;
    if (S == 0) {
        EP_HOST_ASSERT(num_experts % num_ranks == 0);
        v.idx = topk_idx, v.E = (int)num_experts, v.K = K, v.L = v.local_experts = (int)num_experts / W;
        return v;
    }
    EP_HOST_ASSERT_S(num_experts % (W - S) == 0, "num_experts (", num_experts, ") must be a multiple of the ", W - S, " routed-expert ranks");
    EP_HOST_ASSERT_S(K + 1 <= MI_EP_MAX_TOPK, "num_topk + the shared selection must be <= ", MI_EP_MAX_TOPK);
    v.L = (int)num_experts / (W - S), v.E = W * v.L, v.K = K + 1, v.local_experts = rank < S ? 1 : v.L;
    v.idx = at::empty({T, K + 1}, at::dtype(at::kInt).device(topk_idx.device()));
    if (want_weights) v.weights = at::empty({T, K + 1}, at::dtype(at::kFloat).device(topk_idx.device()));
    MI_EP_CHECK(mi_ep_shared_expert_map(topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, weights, T, K, (int)num_experts, W, S, (int)rank,
                                        v.idx.data_ptr<int>(), want_weights ? v.weights.data_ptr<float>() : nullptr, st));
    return v;
}

// ------------------------------------------------------------------------------------------------
// A5  low_latency_dispatch  (reference deep_ep.cpp:850-1012)
// ------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, std::optional<at::Tensor>, at::Tensor, at::Tensor, at::Tensor, std::optional<EventHandle>,
           std::optional<std::function<void()>>>
Buffer::low_latency_dispatch(const at::Tensor &x, const at::Tensor &topk_idx_user, const std::optional<at::Tensor> &,
                             int64_t num_max_dispatch_tokens_per_rank, int64_t num_experts, bool, bool, bool, bool, bool,
                             bool, const std::string &quant_mode_name)
{
    require_available();
    EP_HOST_ASSERT(low_latency_mode);
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous() and x.scalar_type() == at::kBFloat16);
    EP_HOST_ASSERT(num_max_dispatch_tokens_per_rank >= x.size(0));
    EP_HOST_ASSERT(topk_idx_user.dim() == 2 and topk_idx_user.is_contiguous() and topk_idx_user.size(0) == x.size(0));
    EP_HOST_ASSERT(topk_idx_user.scalar_type() == at::kLong or topk_idx_user.scalar_type() == at::kInt);
    const int T = (int)x.size(0), H = (int)x.size(1), K_user = (int)topk_idx_user.size(1);
    const int W = (int)num_ranks, MT = (int)num_max_dispatch_tokens_per_rank, S = (int)shared_expert_rank_num;
    // shared-expert ranks: the kernels see K + 1 selections over W * L expert slots (shared_view); outputs keep the reference's shapes
    const SharedView sv = shared_view(topk_idx_user, nullptr, false, num_experts, cur_stream());
    const at::Tensor &topk_idx = sv.idx;
    const int E = sv.E, K = sv.K, L = sv.L;
    int qm;
    if (quant_mode_name == "int8") qm = MI_EP_QUANT_INT8_NOEPS;
    else if (quant_mode_name == "pertoken_fp8_e4m3") qm = MI_EP_QUANT_FP8_E4M3;
    else {
        const bool not_built = quant_mode_name == "mx_fp8_e4m3" || quant_mode_name == "mx_fp8_e5m2" || quant_mode_name == "mx_fp4_e2m1";
        EP_HOST_ASSERT_S(!not_built, quant_mode_name, " is not supported on this device, please use int8, pertoken_fp8_e4m3 or bf16 instead.");
        EP_HOST_ASSERT(quant_mode_name == "none");
        qm = MI_EP_QUANT_NONE;
    }
    EP_HOST_ASSERT_S(H % 16 == 0 && H <= MI_EP_MAX_HIDDEN, "hidden (", H, ") must be a multiple of 16 and <= ", MI_EP_MAX_HIDDEN);
    EP_HOST_ASSERT_S(E <= 2048 && K <= MI_EP_MAX_TOPK, "num_experts <= 2048 and num_topk <= ", MI_EP_MAX_TOPK);
    const size_t rb = mi_ep_dispatch_row_bytes(H, qm);
    EP_HOST_ASSERT_S((size_t)L * W * MT * rb <= region_bytes, "low-latency window too small: need ", (size_t)L * W * MT * rb,
                     " bytes per region, have ", region_bytes, "; raise DEEPEP_WINDOW_BYTES");
    check_status("low_latency_dispatch");
    ++profile_calls;
    // deep_ep.cpp:866-874: a shared rank receives global_bs / S rows at most, a routed-expert rank global_bs * min(K, L) (the shared
    // selection never lands there)
    const int64_t num_max_tokens = S > 0 && rank < S ? (int64_t)MT * W / S : (int64_t)MT * W * std::min(K_user, L);
    const int64_t max_size = std::max<int64_t>((int64_t)T * K_user, num_max_tokens * 128);   // deep_ep.cpp:875
    auto dev = x.device();
    auto i32 = at::dtype(at::kInt).device(dev);
    const int count_type = get_value_from_env("MOE_EXPERT_TOKEN_NUMS_TYPE", 1);
    hipStream_t st = cur_stream();
    uint64_t *ctr = epoch_ctr(kLLDispatch);
    auto row_peers = peer_family_bases(kLLDispatch);
    // Layout + send in ONE launch (MI_EP_LL_FUSED=0: the two launches): the layout workgroup and the send waves of the same kernel run
    // side by side; a send wave counts its slab position itself in an LDS copy of the routing table (mi_ep_ll_dispatch_layout_send).
    static const bool fused_send = get_value_from_env("MI_EP_LL_FUSED", 1) != 0;
    Layout lay;
    at::Tensor counts_buf;
    bool counts_done = false, tagged_rows = false;
    last_ll_call_was_combine = false;
    last_ll_dispatch_capture = capture_id_of(cur_stream());
    // The TAGGED wire form (rows carry a tag, no count exchange launch) changes what the receivers wait for, so it is chosen from values every
    // rank shares -- the batch bound MT, E, the env -- never from this rank's own T: a rank with T > 1024 beside ranks with T <= 1024 would
    // otherwise send plain rows to receivers that wait for tags.  (MT <= 1024 implies T <= 1024: the one-launch layout + send fits.)
    const bool tagged_form = fused_send && ll_launch_form("MI_EP_LL_FUSED_COUNTS") == 2 && MT <= 1024 && (size_t)16 * E <= 16384 && E % 2 == 0;
    if (fused_send && T <= 1024 && (size_t)16 * E <= 16384 && E % 2 == 0) {
        lay.T = T, lay.K = K, lay.E = E;
        // the five layout tables carved out of ONE allocation (each at::empty costs ~1-2 us of host time in front of the first launch)
        auto pad = [](int64_t n) { return (n + 3) / 4 * 4; };      // every table 16-byte aligned
        at::Tensor lbuf = at::empty({2 * pad(E) + pad(W) + pad((int64_t)T * W) + pad((int64_t)T * K)}, i32);
        int64_t off = 0;
        auto take = [&](int64_t n) { at::Tensor t = lbuf.narrow(0, off, n); off += pad(n); return t; };
        lay.num_tokens_per_expert = take(E);
        lay.num_tokens_per_rank = take(W);
        lay.is_token_in_rank = take((int64_t)T * W).view({T, W});
        lay.send_token_idx_small = take((int64_t)T * K).view({T, K});
        lay.send_data_offset = take(E);
        // ... and, opt-in (MI_EP_LL_FUSED_COUNTS=1), the count exchange in the same launch: its last workgroup to finish posts and
        // collects the counts, two launches per dispatch instead of three.  Built, bit-exact, and SLOWER on one GPU: 25.4 against 22.7 us
        // per call (pair in a graph of ten: 31.7 against 27.0) -- every workgroup pays a drain + a device-scope arrival, the layout
        // workgroup a release, the rows go through the caches to HBM instead of waiting in L2 for the packing launch; the kernel boundary
        // it saves costs less than that (MI355X_MICROARCH.md: ~1.5 us).  Left off, like the combine side of the same idea (combine_push_rows).
        // MI_EP_LL_FUSED_COUNTS=2: TWO launches with nothing between them -- the layout workgroup posts the counts at the head of the send launch,
        // the send waves tag their rows (meta word written behind the drained payload), and the packing launch collects counts and rows itself
        // (mi_ep_ll_dispatch_layout_send_tagged / mi_ep_ll_wait_pack): no workgroup of the send launch waits for another.
        // One GPU, same box, alternating (bench.py's low-latency section, two runs each): queued dispatch 20.0-22.3 -> 17.6-18.2 us, replayed
        // 24.6-24.8 -> 23.4-23.6; with the two-launch combine (combine_push_rows) the dispatch + combine pair in a graph of ten
        // 26.4-26.7 -> 22.7-22.8 us.  Default; 0 = three launches, 1 = the last-arriver tail below.
        // (ranks that share a GPU default to three launches: deep_ep.hpp, get_local_device_bus_id)
        const int ll_form = ll_launch_form("MI_EP_LL_FUSED_COUNTS");
        const bool fused_counts = ll_form == 1;
        if (tagged_form) {
            auto cnt_peers0 = peer_ptrs((size_t)kOffLLCounts);
            ProfScope ps_(this, "ll_dispatch_layout_send", st);
            MI_EP_CHECK(mi_ep_ll_dispatch_layout_send_tagged(
                x.data_ptr(), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, T, K, H, E, W, (int)rank, MT, qm, row_peers.data(), ctr,
                region_bytes, lay.num_tokens_per_rank.data_ptr<int>(), lay.num_tokens_per_expert.data_ptr<int>(),
                lay.is_token_in_rank.data_ptr<int>(), lay.send_token_idx_small.data_ptr<int>(), lay.send_data_offset.data_ptr<int>(),
                (uint64_t *const *)cnt_peers0.data(), (size_t)kLLCountsParityBytes, (uint64_t *)(window + kOffEpochs + 1024 + 136), st));
            tagged_rows = true;
        } else if (fused_counts) {
            counts_buf = at::empty({(int64_t)L * W + 2 * (int64_t)L + 2}, i32);      // ep_recv_count [L*W] | packed_recv_count [L] (int64, 8-byte aligned)
            auto cnt_peers0 = peer_ptrs((size_t)kOffLLCounts);
            const int64_t off64 = ((int64_t)L * W + 1) / 2 * 2;
            ProfScope ps_(this, "ll_dispatch_layout_send_counts", st);
            MI_EP_CHECK(mi_ep_ll_dispatch_layout_send_counts(
                x.data_ptr(), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, T, K, H, E, W, (int)rank, MT, qm, row_peers.data(), ctr,
                region_bytes, lay.num_tokens_per_rank.data_ptr<int>(), lay.num_tokens_per_expert.data_ptr<int>(),
                lay.is_token_in_rank.data_ptr<int>(), lay.send_token_idx_small.data_ptr<int>(), lay.send_data_offset.data_ptr<int>(),
                (uint64_t *const *)cnt_peers0.data(), (const uint64_t *)(window + kOffLLCounts), (size_t)kLLCountsParityBytes, count_type,
                counts_buf.data_ptr<int>(), (int64_t *)(counts_buf.data_ptr<int>() + off64), arrive_word(), status_dev, timeout_ms, st));
            counts_done = true;
        } else {
        ProfScope ps_(this, "ll_dispatch_layout_send", st);
        MI_EP_CHECK(mi_ep_ll_dispatch_layout_send(x.data_ptr(), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, T, K, H, E, W, (int)rank,
                                                  MT, qm, row_peers.data(), ctr, region_bytes, lay.num_tokens_per_rank.data_ptr<int>(),
                                                  lay.num_tokens_per_expert.data_ptr<int>(), lay.is_token_in_rank.data_ptr<int>(),
                                                  lay.send_token_idx_small.data_ptr<int>(), lay.send_data_offset.data_ptr<int>(), st));
        }
    } else {
        lay = run_layout(topk_idx, E);
        ProfScope ps_(this, "ll_dispatch_send", st);
        MI_EP_CHECK(mi_ep_ll_dispatch_send(x.data_ptr(), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt,
                                           lay.send_token_idx_small.data_ptr<int>(), T, K, H, E, W, (int)rank, MT, qm, row_peers.data(), ctr,
                                           region_bytes, st));
    }
    // The outputs are allocated AFTER the first launch, which does not touch them: a lone decode-step call is host-bound until its first
    // kernel is queued (the GPU idles while the host prepares), and five allocations are ~6 us of that.
    at::Tensor packed_recv_x, packed_recv_x_scales;
    if (qm == MI_EP_QUANT_NONE) {
        packed_recv_x = at::empty({num_max_tokens, H}, at::dtype(at::kBFloat16).device(dev));
        packed_recv_x_scales = at::empty({1}, at::dtype(at::kFloat).device(dev));
    } else {
        packed_recv_x = at::empty({num_max_tokens, H}, at::dtype(payload_dtype(qm)).device(dev));
        packed_recv_x_scales = at::empty({num_max_tokens}, at::dtype(at::kFloat).device(dev));
    }
    auto expand_idx = at::empty({max_size}, i32);
    at::Tensor ep_recv_count, packed_recv_count;
    if (counts_done) {                                          // (views of the buffer the first launch filled)
        const int64_t off64 = ((int64_t)L * W + 1) / 2 * 2;
        ep_recv_count = counts_buf.narrow(0, 0, (int64_t)L * W);
        packed_recv_count = counts_buf.narrow(0, off64, 2 * (int64_t)L).view(at::kLong);
    } else {
        ep_recv_count = at::empty({(int64_t)L * W}, i32);
        packed_recv_count = at::empty({L}, at::dtype(at::kLong).device(dev));
    }
    auto cnt_peers = peer_ptrs((size_t)kOffLLCounts);
    // rows the output tensors hold (and src_info / 3): the packing kernel never writes past them, whatever the counts say
    const int rows_capacity = (int)std::min<int64_t>(num_max_tokens, max_size / 3);
    if (tagged_rows) {
        ProfScope ps_(this, "ll_dispatch_recv", st);
        MI_EP_CHECK(mi_ep_ll_wait_pack(family_base(kLLDispatch), (const uint64_t *)(window + kOffLLCounts), (size_t)kLLCountsParityBytes, W, L, MT, H,
                                       qm, count_type, packed_recv_x.data_ptr(),
                                       qm == MI_EP_QUANT_NONE ? nullptr : packed_recv_x_scales.data_ptr<float>(),
                                       (int64_t *)packed_recv_count.data_ptr(), expand_idx.data_ptr<int>(), ep_recv_count.data_ptr<int>(),
                                       rows_capacity, (const uint64_t *)(window + kOffEpochs + 1024 + 136), ctr, region_bytes, status_dev,
                                       timeout_ms, ranks_share_device ? 64 : 0, st));
    } else if (counts_done) {
        ProfScope ps_(this, "ll_dispatch_recv", st);
        MI_EP_CHECK(mi_ep_ll_pack(family_base(kLLDispatch), ep_recv_count.data_ptr<int>(), W, L, MT, H, qm, packed_recv_x.data_ptr(),
                                  qm == MI_EP_QUANT_NONE ? nullptr : packed_recv_x_scales.data_ptr<float>(), expand_idx.data_ptr<int>(),
                                  rows_capacity, ctr, region_bytes, st));
    } else
    { ProfScope ps_(this, "ll_dispatch_recv", st);
      MI_EP_CHECK(mi_ep_ll_post_recv((uint64_t *const *)cnt_peers.data(), lay.num_tokens_per_expert.data_ptr<int>(), (int)rank,
                                     family_base(kLLDispatch), (const uint64_t *)(window + kOffLLCounts), 0, W, L, MT, H, qm, count_type,
                                     packed_recv_x.data_ptr(), qm == MI_EP_QUANT_NONE ? nullptr : packed_recv_x_scales.data_ptr<float>(),
                                     (int64_t *)packed_recv_count.data_ptr(), expand_idx.data_ptr<int>(), ep_recv_count.data_ptr<int>(),
                                     rows_capacity, ctr, region_bytes, (size_t)kLLCountsParityBytes, status_dev, timeout_ms, st)); }
    real_max_bs = std::max<int64_t>(real_max_bs, MT);
    if (sv.local_experts != L) {      // a shared rank: its single local expert is slot 0 (deep_ep.cpp:869-871: num_local_experts = 1)
        packed_recv_count = packed_recv_count.narrow(0, 0, sv.local_experts);
        ep_recv_count = ep_recv_count.narrow(0, 0, (int64_t)sv.local_experts * W);
    }
    return {packed_recv_x, packed_recv_x_scales, packed_recv_count, expand_idx, ep_recv_count, std::nullopt,
            std::function<void()>([] {})};
}

// ------------------------------------------------------------------------------------------------
// A6  low_latency_combine  (reference deep_ep.cpp:1014-1087)
// ------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, std::optional<EventHandle>, std::optional<std::function<void()>>>
Buffer::low_latency_combine(const at::Tensor &x, const at::Tensor &topk_idx_user, const at::Tensor &topk_weights_user,
                            const at::Tensor &src_info, const at::Tensor &layout_range,
                            int64_t num_max_dispatch_tokens_per_rank, int64_t num_experts, const at::Tensor &, bool, bool,
                            bool, const std::optional<at::Tensor> &)
{
    require_available();
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous() and x.scalar_type() == at::kBFloat16);
    EP_HOST_ASSERT(num_max_dispatch_tokens_per_rank >= topk_idx_user.size(0));
    EP_HOST_ASSERT(topk_idx_user.dim() == 2 and topk_idx_user.is_contiguous());
    EP_HOST_ASSERT(topk_idx_user.scalar_type() == at::kLong or topk_idx_user.scalar_type() == at::kInt);
    EP_HOST_ASSERT(topk_weights_user.dim() == 2 and topk_weights_user.is_contiguous() and topk_weights_user.scalar_type() == at::kFloat);
    EP_HOST_ASSERT(topk_weights_user.size(0) == topk_idx_user.size(0) and topk_weights_user.size(1) == topk_idx_user.size(1));
    EP_HOST_ASSERT(src_info.scalar_type() == at::kInt and layout_range.scalar_type() == at::kInt);
    // shared-expert ranks: K + 1 slots per token, the last one the shared expert's row with weight 1 (moe_distribute_combine_v2.h:1219-1235)
    const SharedView sv = shared_view(topk_idx_user, topk_weights_user.data_ptr<float>(), true, num_experts, cur_stream());
    const at::Tensor &topk_idx = sv.idx;
    const at::Tensor &topk_weights = shared_expert_rank_num > 0 ? sv.weights : topk_weights_user;
    const int K = sv.K, H = (int)x.size(1);
    const int W = (int)num_ranks, E = sv.E;
    const size_t cb = mi_ep_combine_row_bytes(H);
    EP_HOST_ASSERT_S((size_t)num_max_dispatch_tokens_per_rank * K * cb <= region_bytes, "combine window too small; raise DEEPEP_WINDOW_BYTES");
    check_status("low_latency_combine");
    hipStream_t st = cur_stream();
    auto dst_peers = peer_family_bases(kCombine);
    // valid packed rows = layout_range[L*W-1], read on device
    auto local_row = combine_local_rows(topk_idx);
    int signalled = 0;
    // (the per-row flag form needs a flag word per slot on EVERY rank: decided from num_max_dispatch_tokens_per_rank, which the ranks share)
    // (the last words of each half are the start-up self-test's: self_test_in_launch)
    const bool flag_words_ok = (int64_t)num_max_dispatch_tokens_per_rank * K <=
                               kRowFlagsParityBytes / 4 - (int64_t)mi_ep_selftest_inlaunch_flag_words(MI_EP_MAX_RANKS);
    combine_push_rows(x, src_info.data_ptr<int>(), layout_range.data_ptr<int>() + (layout_range.numel() - 1),
                      (int)std::min<int64_t>(x.size(0), src_info.numel() / 3), H, K, local_row, "ll_combine_push", st, signalled, flag_words_ok);
    // the `out=` argument is accepted and a fresh tensor is returned, as in the reference (deep_ep.cpp:1057)
    return {combine_finish(topk_idx, topk_weights.data_ptr<float>(), H, E, x.options(), "ll_combine_reduce", st, x, local_row, signalled),
            std::nullopt, std::function<void()>([] {})};
}

// EP = 1: receive rows recorded by the dispatch, keyed by the handle's recv_src_idx tensor (held, so its address cannot be recycled while the
// entry lives); a few exchanges may be in flight between their dispatch and their combine (two-batch overlap)
void Buffer::remember_local_rows(const at::Tensor &src_idx, const at::Tensor &rows, int T, int K)
{
    if (local_row_stash.size() >= 8) local_row_stash.erase(local_row_stash.begin());
    local_row_stash.push_back(LocalRowEntry{src_idx, rows, T, K});
}

at::Tensor Buffer::recall_local_rows(const at::Tensor &src_idx, int T, int K) const
{
    for (auto it = local_row_stash.rbegin(); it != local_row_stash.rend(); ++it)
        if (it->src_idx.data_ptr() == src_idx.data_ptr() && it->src_idx.numel() >= src_idx.numel() && it->T == T && it->K == K)
            return it->rows;
    return at::Tensor();
}

// second half of a combine: "my rows are pushed" to every owner, wait for every expert rank (ONE single-wave launch, which also
// completes the family's device-resident call counter), then the weighted sum over the K slots of every token
// One int32 per (token, selection) of this rank: the row of the expert output that holds it, for the selections this rank's own
// experts served (written by the combine push, read by the reduce).  MI_EP_COMBINE_LOCAL=0 sends those rows through the window
// like everyone else's.
at::Tensor Buffer::combine_local_rows(const at::Tensor &topk_idx) const
{
    if (!combine_local_rows_enabled || topk_idx.numel() == 0) return at::Tensor();
    return at::empty({topk_idx.numel()}, at::dtype(at::kInt).device(topk_idx.device()));
}

// Device words the fused push counts its workgroups in at (control area, zeroed at creation, re-armed by the kernel): dealt from a ring of
// 16, so that calls of one Buffer that overlap on two streams (two-batch overlap) never share one; a captured call keeps its word, and
// the replays of one graph serialise on its stream.
std::string Buffer::get_local_device_bus_id() const
{
    char id[64] = {0};
    if (hipDeviceGetPCIBusId(id, (int)sizeof(id) - 1, device_id) != hipSuccess) return std::string("device-") + std::to_string(device_id);
    return std::string(id);
}

int Buffer::ll_launch_form(const char *env_name) const
{
    const char *e = getenv(env_name);
    const int form = e && *e ? atoi(e) : (ranks_share_device ? 0 : 2);
    // the in-launch hand-off did not pass its start-up test on some rank: three launches, whatever was asked for
    return form == 2 && !two_launch_forms_ok ? 0 : form;
}

uint32_t *Buffer::arrive_word() { return (uint32_t *)(window + kOffEpochs + 1024) + 2 * (arrive_calls++ % 16); }

void Buffer::combine_push_rows(const at::Tensor &x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int H, int K,
                               const at::Tensor &local_row, const char *name, hipStream_t st, int &signalled, bool may_flag_rows)
{
    // MI_EP_COMBINE_FUSED: how the owners learn that their rows have arrived.
    //   2 (default where every rank owns its GPU -- ranks sharing one keep 0, deep_ep.hpp; low-latency combine only, may_flag_rows): TWO launches without any exchange between them -- every pushed row raises its own
    //     flag word at its owner, the reduce waits per selection (mi_ep_combine_push_flagged / mi_ep_combine_reduce_flagged; the reference's
    //     per-token wait, moe_distribute_combine_v2.h:952-1002).  One GPU, same box, alternating: a lone combine 13.4-13.6 -> 11.1-11.5 us,
    //     replayed 20.4 -> 18.2-18.5, the dispatch + combine pair in a graph of ten 26.6-27.0 -> 24.6-24.7 us.
    //   0: three launches (push, the single-wave "rows pushed" signal + wait, reduce) -- what every other combine uses.
    //   1: the signal + wait as the TAIL of the push (last workgroup to arrive): a lone low-latency combine 13.1-13.3 us against 13.5-16.0 for
    //     the three launches, but the pair inside a captured graph of ten 28.7 against 27.2 us (256 workgroups arriving at one word and a
    //     tail that waits inside the push cost more than the ~1.3 us boundary they replace); normal-mode step unchanged.
    int fused_form = ll_launch_form("MI_EP_COMBINE_FUSED");
    // see last_ll_call_was_combine (deep_ep.hpp); and a combine recorded into a graph WITHOUT its dispatch in the same capture would be
    // replayed combine after combine: three launches as well
    if (fused_form == 2 && may_flag_rows && (last_ll_call_was_combine || capture_id_of(st) != last_ll_dispatch_capture)) fused_form = 0;
    if (may_flag_rows) last_ll_call_was_combine = true;
    const bool fused = fused_form == 1;
    const int W = (int)num_ranks;
    auto dst_peers = peer_family_bases(kCombine);
    int32_t *lr = local_row.defined() ? local_row.data_ptr<int>() : nullptr;
    ProfScope ps_(this, name, st);
    if (fused_form == 2 && may_flag_rows) {
        auto row_flag_peers = peer_ptrs((size_t)kOffRowFlags);
        MI_EP_CHECK(mi_ep_combine_push_flagged(x.data_ptr(), src_idx, total_rows_dev, rows_hint, H, K, dst_peers.data(), W, region_bytes,
                                               epoch_ctr(kCombine), region_bytes, (int)rank, lr, (uint32_t *const *)row_flag_peers.data(),
                                               (size_t)kRowFlagsParityBytes, (uint64_t *)(window + kOffEpochs + 1024 + 128), st));
        signalled = 2;
    } else if (fused) {
        auto flag_peers = peer_ptrs((size_t)(kOffFlags + kFlagCombine * kFlagGroupSlots * 8));
        MI_EP_CHECK(mi_ep_combine_push_signal_wait(x.data_ptr(), src_idx, total_rows_dev, rows_hint, H, K, dst_peers.data(), W, region_bytes,
                                                   epoch_ctr(kCombine), region_bytes, (int)rank, lr, (uint64_t *const *)flag_peers.data(),
                                                   (const uint64_t *)(window + kOffFlags + kFlagCombine * kFlagGroupSlots * 8), arrive_word(),
                                                   status_dev, timeout_ms, st));
        signalled = 1;
    } else {
        MI_EP_CHECK(mi_ep_combine_push(x.data_ptr(), src_idx, total_rows_dev, rows_hint, H, K, dst_peers.data(), W, region_bytes,
                                       epoch_ctr(kCombine), region_bytes, (int)rank, lr, st));
    }
}

at::Tensor Buffer::combine_finish(const at::Tensor &topk_idx, const float *topk_weights, int H, int E, const at::TensorOptions &opts,
                                  const char *reduce_name, hipStream_t st, const at::Tensor &x_local, const at::Tensor &local_row, int signalled)
{
    const int T = (int)topk_idx.size(0), K = (int)topk_idx.size(1), W = (int)num_ranks;
    auto flag_peers = peer_ptrs((size_t)(kOffFlags + kFlagCombine * kFlagGroupSlots * 8));
    if (signalled == 2) {                        // rows with flags: the reduce waits for them itself and completes the call counter
        auto combined_x = at::empty({T, H}, opts);
        const bool use_local = local_row.defined() && x_local.defined() && x_local.size(0) > 0;
        ProfScope ps_(this, reduce_name, st);
        MI_EP_CHECK(mi_ep_combine_reduce_flagged(family_base(kCombine), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, topk_weights, T, K,
                                                 H, E, combined_x.data_ptr(), epoch_ctr(kCombine), region_bytes,
                                                 use_local ? x_local.data_ptr() : nullptr, use_local ? local_row.data_ptr<int>() : nullptr,
                                                 use_local ? (int)x_local.size(0) : 0, (int)rank, W,
                                                 (const uint32_t *)(window + kOffRowFlags), (size_t)kRowFlagsParityBytes,
                                                 (const uint64_t *)(window + kOffEpochs + 1024 + 128), status_dev, timeout_ms,
                                                 ranks_share_device ? 64 : 0, st));
        return combined_x;
    }
    if (!signalled) { ProfScope ps_(this, "combine_signal_wait", st);
      MI_EP_CHECK(mi_ep_signal_wait((uint64_t *const *)flag_peers.data(),
                                    (const uint64_t *)(window + kOffFlags + kFlagCombine * kFlagGroupSlots * 8), W, (int)rank, 0,
                                    epoch_ctr(kCombine), status_dev, timeout_ms, st)); }
    auto combined_x = at::empty({T, H}, opts);
    const bool use_local = local_row.defined() && x_local.defined() && x_local.size(0) > 0;
    { ProfScope ps_(this, reduce_name, st);
      MI_EP_CHECK(mi_ep_combine_reduce(family_base(kCombine), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt, topk_weights,
                                       nullptr, nullptr, T, K, H, E, combined_x.data_ptr(), epoch_ctr(kCombine), region_bytes,
                                       use_local ? x_local.data_ptr() : nullptr, use_local ? local_row.data_ptr<int>() : nullptr,
                                       use_local ? (int)x_local.size(0) : 0, (int)rank, W, st)); }
    return combined_x;
}

// the nine int32 tables of one notify exchange carved out of ONE allocation (each at::empty costs ~1-2 us of host time)
Buffer::NotifyTables Buffer::alloc_notify_tables(int W, int E, int L, const at::TensorOptions &i32)
{
    const int64_t n_cnt = (int64_t)W * (E + 1);
    auto pad = [](int64_t n) { return (n + 3) / 4 * 4; };          // keep every table 16-byte aligned
    const int64_t total = pad(n_cnt) + 5 * pad(E) + 2 * pad(L) + 2 * 4;
    at::Tensor buf = at::empty({total}, i32);
    int64_t off = 0;
    auto take = [&](int64_t n) { at::Tensor t = buf.narrow(0, off, n); off += pad(n); return t; };
    NotifyTables nt;
    nt.cnt = take(n_cnt).view({W, E + 1});
    nt.recv_count = take(E), nt.recv_offset = take(E), nt.srcrank_in_expert_offset = take(E);
    nt.r_in_srcrank_offset = take(E), nt.pull_offset = take(E);
    nt.recv_tokens_per_expert = take(L), nt.expert_global_offset = take(L);
    nt.total_recv_token = take(1), nt.max_bs = take(1);
    return nt;
}

void Buffer::internode_unsupported() const
{
    throw EPException("Assertion", __FILE__, __LINE__,
                      "internode (multi-node RDMA) dispatch/combine is out of scope: one MI355X xGMI node is a single rdma rank");
}

// ------------------------------------------------------------------------------------------------
// A8  fused_deep_moe  (reference deep_ep.cpp:1089-1113,1214-1233; kernel ops/op_kernel/fused_deep_moe.h)
// dispatch (INT8 per token, low-latency layout) -> grouped GEMM1 + dequant + SwiGLU -> per-row requant -> grouped GEMM2 +
// dequant -> weighted combine.  The reference runs this as ONE Ascend MIX kernel; here it is a chain of launches on the
// caller's stream sharing buffers (no host synchronisation anywhere), each stage a HIP kernel of include/mi_ep.h.
// Weight layout on MI355X: gmm1_permuted_weight int8 [L, 2I, H] and gmm2_weight int8 [L, H, I] (output channel major,
// K contiguous; the Ascend NZ fractal format does not exist here).  The reference's logical shapes [L, H, 2I] / [L, I, H]
// are accepted and transposed on the fly.
// ------------------------------------------------------------------------------------------------
// Weights are layer constants: any re-layout (transpose to K-contiguous, fusion-tile row permutation) is done once per
// (storage, version) and kept, like the reference's one-off npu_format_cast to NZ (test_fused_deep_moe.py:63-119).
at::Tensor Buffer::prepared_weight(const at::Tensor &w, int kind, const std::function<at::Tensor()> &make)
{
    const WeightKey key{w.data_ptr(), kind};
    auto it = weight_cache_.find(key);
    // Inference tensors have no version counter (version -1): the entry then hangs on (address, held storage, numel) alone;
    // weights are layer constants, a caller that rewrites an inference-mode weight in place calls clear_weight_cache().
    const int64_t ver = Layout::version_of(w);
    if (it != weight_cache_.end() && it->second.version == ver && it->second.numel == w.numel()) return it->second.t;
    if (weight_cache_.size() >= 512) weight_cache_.clear();
    at::Tensor t = make();
    weight_cache_[key] = WeightEntry{ver, w.numel(), w, t};
    return t;
}

std::vector<at::Tensor> Buffer::fused_core(const at::Tensor &x, const at::Tensor &expert_ids, const at::Tensor &w1,
                                           const at::Tensor &s1, const at::Tensor &w2, const at::Tensor &s2,
                                           const at::Tensor &topk_weights, int64_t num_max_dispatch_tokens_per_rank,
                                           int64_t num_experts)
{
    const int W = (int)num_ranks, H = (int)x.size(1), T = (int)x.size(0), K_user = (int)expert_ids.size(1), S = (int)shared_expert_rank_num;
    const int N1 = (int)w1.size(1), I = N1 / 2;
    hipStream_t st = cur_stream();
    EP_HOST_ASSERT(topk_weights.size(0) == T and topk_weights.size(1) == K_user);
    // shared-expert ranks (deep_ep.cpp:1219-1220): a shared rank runs ONE expert over every token of its W / S sources, the routed ranks
    // their L experts; the kernels see K + 1 selections over W * L expert slots (shared_view), Lw = this rank's local experts
    const SharedView sv = shared_view(expert_ids, topk_weights.data_ptr<float>(), true, num_experts, st);
    const int E = sv.E, L = sv.L, K = sv.K, Lw = sv.local_experts;
    const at::Tensor &ids = sv.idx;
    const float *weights = S > 0 ? sv.weights.data_ptr<float>() : topk_weights.data_ptr<float>();
    EP_HOST_ASSERT(w1.is_contiguous() and w2.is_contiguous() and w1.size(0) == Lw and w2.size(0) == Lw);
    EP_HOST_ASSERT(w1.size(2) == H and w2.size(1) == H and w2.size(2) == I);
    EP_HOST_ASSERT(s1.numel() == (int64_t)Lw * N1 and s2.numel() == (int64_t)Lw * H);
    EP_HOST_ASSERT_S(H % 128 == 0 && I % 128 == 0, "hidden (", H, ") and intermediate (", I, ") must be multiples of 128");
    const int64_t MT = num_max_dispatch_tokens_per_rank;
    auto dev = x.device();

    // Dispatch leg.  Decode-size batches take the low-latency slabs (rows straight into the destination's (expert, source)
    // slab, 4 launches).  Prefill-size batches take the normal-mode exchange without its host sync: one staged row per TOKEN
    // instead of one per (token, k) -- 31 MB instead of 235 MB of staging at 4096 tokens -- and a window of T rows per source
    // instead of L x W x max_tokens slabs (7.5 GB at EP = 8 x 4096 tokens, which no window holds).  Both produce the same packed
    // rows in (local expert, source rank) order, the same triples and the same inclusive counts.
    at::Tensor rx, rs, src_info, layout_range, a_rows;
    const void *a_base = nullptr;
    const size_t ll_bytes = (size_t)L * W * MT * mi_ep_dispatch_row_bytes(H, MI_EP_QUANT_INT8_NOEPS);
    if (MT <= 512 && ll_bytes <= region_bytes) {
        std::optional<at::Tensor> none;
        auto disp = low_latency_dispatch(x, expert_ids, none, MT, num_experts, true, false, false, false, false, false, "int8");
        rx = std::get<0>(disp), rs = *std::get<1>(disp), src_info = std::get<3>(disp), layout_range = std::get<4>(disp);
    } else {
        check_status("fused_deep_moe");
        ++profile_calls;
        const Layout lay = run_layout(ids, E);
        DispatchExchange ex = dispatch_exchange(x, ids, lay, E, MI_EP_QUANT_INT8_NOEPS, false, nullptr, st);
        const int64_t rows_cap = std::max<int64_t>(1, S > 0 && rank < S ? MT * W / S : MT * W * std::min(K_user, L));   // worst case (deep_ep.cpp:866-874)
        // GEMM1 multiplies the staged rows WHERE THEY ARE (one row per token, shared by its K selections) when every source is local memory
        // -- own region at one rank, the source slabs of the own window under the push transport --: a table of row offsets instead of the
        // K-fold gathered copy.  MI_EP_FUSED_GATHER=0 / set_fused_rows_in_place(false), remote sources (pull transport) or a window wider than 32 bits: gather as before.
        if (fused_rows_in_place && (ex.push || W == 1)) {
            a_rows = at::empty({rows_cap}, at::dtype(at::kInt).device(dev));
            rs = at::empty({rows_cap}, at::dtype(at::kFloat).device(dev));
            src_info = at::empty({rows_cap * 3}, at::dtype(at::kInt).device(dev));
            ProfScope ps_(this, "dispatch_resolve_rows", st);
            const int rc = mi_ep_dispatch_resolve_rows((const void *const *)ex.src_bases.data(), ex.nt.recv_count.data_ptr<int>(),
                                                       ex.nt.pull_offset.data_ptr<int>(), W, L, H, K, MI_EP_QUANT_INT8_NOEPS, (int)rows_cap,
                                                       ex.slab_bytes, &a_base, (uint32_t *)a_rows.data_ptr<int>(), rs.data_ptr<float>(),
                                                       src_info.data_ptr<int>(), epoch_ctr(kDispatch), region_bytes, st);
            if (rc == MI_EP_ESIZE) a_rows = at::Tensor(), a_base = nullptr;
            else MI_EP_CHECK(rc);
        }
        if (!a_rows.defined()) dispatch_pull(ex, H, K, L, MI_EP_QUANT_INT8_NOEPS, rows_cap, x.options(), rx, rs, src_info, st);
        layout_range = Lw != L ? ex.nt.recv_count.narrow(0, 0, (int64_t)Lw * W) : ex.nt.recv_count;
        real_max_bs = std::max<int64_t>(real_max_bs, MT);
    }
    const int M = (int)(a_rows.defined() ? a_rows.size(0) : rx.size(0));
    // opt-in (MI_EP_FUSED_REQUANT=1 / set_fused_requant(true)): GEMM1 requantises its rows in its own epilogue (the reference's structure,
    // block_epilogue_per_token_dequant_swiglu.h:250-269): no fp32 [M, I] intermediate, no rowquant launch; same bits, not faster (deep_ep.hpp)
    const bool requant_in_gemm1 = fused_requant && N1 % 256 == 0;
    at::Tensor v = requant_in_gemm1 ? at::Tensor() : at::empty({M, I}, at::dtype(at::kFloat).device(dev));
    at::Tensor q2 = at::empty({M, I}, at::dtype(at::kChar).device(dev));
    at::Tensor sc2 = at::empty({M}, at::dtype(at::kFloat).device(dev));
    const int32_t *cum = layout_range.data_ptr<int>();
    // expected rows per local expert under balanced routing (all ranks send about T tokens x K): picks the GEMM tile shape
    const int rows_hint = S > 0 && rank < S ? std::max(1, T * (W / S))
                                            : (int)std::max<int64_t>(1, (int64_t)T * K_user * W / std::max<int64_t>(1, num_experts));
    if (requant_in_gemm1) {
        const int64_t words = (int64_t)mi_ep_moe_requant_words(M, Lw);
        at::Tensor rq = at::empty({words}, at::dtype(at::kInt).device(dev));
        HIP_CHECK(hipMemsetAsync(rq.data_ptr(), 0, (size_t)words * 4, st));
        ProfScope ps_(this, "moe_gemm1_swiglu_quant", st);
        MI_EP_CHECK(mi_ep_moe_gemm1_swiglu_quant(a_rows.defined() ? a_base : rx.data_ptr(),
                                                 a_rows.defined() ? (const uint32_t *)a_rows.data_ptr<int>() : nullptr, rs.data_ptr<float>(),
                                                 (const int8_t *)w1.data_ptr(), s1.data_ptr<float>(), cum, W, Lw, M, H, N1, (int8_t *)q2.data_ptr(),
                                                 sc2.data_ptr<float>(), (uint32_t *)rq.data_ptr<int>(), gemm_xcds, status_dev, timeout_ms, rows_hint,
                                                 st));
    } else {
    { ProfScope ps_(this, "moe_gemm1_swiglu", st);
      if (a_rows.defined())
          MI_EP_CHECK(mi_ep_moe_gemm1_swiglu_rows(a_base, (const uint32_t *)a_rows.data_ptr<int>(), rs.data_ptr<float>(), (const int8_t *)w1.data_ptr(),
                                                  s1.data_ptr<float>(), cum, W, Lw, M, H, N1, v.data_ptr<float>(), rows_hint, st));
      else
          MI_EP_CHECK(mi_ep_moe_gemm1_swiglu((const int8_t *)rx.data_ptr(), rs.data_ptr<float>(), (const int8_t *)w1.data_ptr(),
                                             s1.data_ptr<float>(), cum, W, Lw, M, H, N1, v.data_ptr<float>(), rows_hint, st)); }
    { ProfScope ps_(this, "moe_rowquant", st);
      MI_EP_CHECK(mi_ep_moe_rowquant(v.data_ptr<float>(), cum + (Lw * W - 1), M, I, (int8_t *)q2.data_ptr(), sc2.data_ptr<float>(), st)); }
    }
    // GEMM2 writes every bf16 row straight into its owner's combine slot (the push of low_latency_combine fused into the GEMM
    // epilogue: no dense [M, H] intermediate, one pass over 2*M*H bytes less), then the usual signal / wait / weighted sum
    const size_t cb = mi_ep_combine_row_bytes(H);
    EP_HOST_ASSERT_S((size_t)MT * K * cb <= region_bytes, "combine window too small; raise DEEPEP_WINDOW_BYTES");
    auto dst_peers = peer_family_bases(kCombine);
    { ProfScope ps_(this, "moe_gemm2_push", st);
      MI_EP_CHECK(mi_ep_moe_gemm2_push((const int8_t *)q2.data_ptr(), sc2.data_ptr<float>(), (const int8_t *)w2.data_ptr(),
                                       s2.data_ptr<float>(), cum, W, Lw, M, I, H, src_info.data_ptr<int>(), K, dst_peers.data(), W,
                                       region_bytes, epoch_ctr(kCombine), region_bytes, rows_hint, st)); }
    at::Tensor combined = combine_finish(ids, weights, H, E, x.options(), "ll_combine_reduce", st);
    return {combined, layout_range};
}

static void fused_common_checks(const at::Tensor &x, const at::Tensor &expert_ids, const at::Tensor &w1, const at::Tensor &w2,
                                int64_t quant_mode, const std::optional<at::Tensor> &expert_scales, const char *what)
{
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous() and x.scalar_type() == at::kBFloat16);
    EP_HOST_ASSERT(expert_ids.dim() == 2 and expert_ids.is_contiguous() and expert_ids.size(0) == x.size(0));
    EP_HOST_ASSERT(expert_ids.scalar_type() == at::kLong or expert_ids.scalar_type() == at::kInt);
    EP_HOST_ASSERT_S(quant_mode == 1, what, ": only quant_mode=1 (INT8 weights) is implemented on this device");
    EP_HOST_ASSERT(expert_scales.has_value() and expert_scales->dim() == 2);
    EP_HOST_ASSERT(w1.dim() == 3 and w2.dim() == 3);
    EP_HOST_ASSERT_S(w1.scalar_type() == at::kChar and w2.scalar_type() == at::kChar, what, ": INT8 weights required");
}

std::vector<at::Tensor> Buffer::fused_deep_moe(const at::Tensor &x, const at::Tensor &expert_ids,
                                               const at::Tensor &gmm1_permuted_weight,
                                               const at::Tensor &gmm1_permuted_weight_scale, const at::Tensor &gmm2_weight,
                                               const at::Tensor &gmm2_weight_scale,
                                               const std::optional<at::Tensor> &expert_scales_optional,
                                               int64_t num_max_dispatch_tokens_per_rank, int64_t num_experts,
                                               int64_t quant_mode, bool)
{
    require_available();
    fused_common_checks(x, expert_ids, gmm1_permuted_weight, gmm2_weight, quant_mode, expert_scales_optional, "fused_deep_moe");
    const int W = (int)num_ranks, S = (int)shared_expert_rank_num, H = (int)x.size(1);
    const int L = S > 0 ? (rank < S ? 1 : (int)num_experts / (W - S)) : (int)num_experts / W;      // deep_ep.cpp:1219-1220
    EP_HOST_ASSERT(gmm1_permuted_weight.size(0) == L and gmm2_weight.size(0) == L);
    at::Tensor w1 = gmm1_permuted_weight, w2 = gmm2_weight;
    if (w1.size(2) != H) {                       // reference logical shape [L, H, 2I]
        EP_HOST_ASSERT(w1.size(1) == H);
        w1 = prepared_weight(gmm1_permuted_weight, 0, [&] { return gmm1_permuted_weight.transpose(1, 2).contiguous(); });
    }
    const int N1 = (int)w1.size(1), I = N1 / 2;
    if (w2.size(1) != H) {                       // reference logical shape [L, I, H]
        EP_HOST_ASSERT(w2.size(2) == H and w2.size(1) == I);
        w2 = prepared_weight(gmm2_weight, 1, [&] { return gmm2_weight.transpose(1, 2).contiguous(); });
    }
    at::Tensor s1 = gmm1_permuted_weight_scale.to(at::kFloat).reshape({L, N1}).contiguous();
    at::Tensor s2 = gmm2_weight_scale.to(at::kFloat).reshape({L, H}).contiguous();
    at::Tensor topk_weights = expert_scales_optional->to(at::kFloat).contiguous();
    return fused_core(x, expert_ids, w1, s1, w2, s2, topk_weights, num_max_dispatch_tokens_per_rank, num_experts);
}

// N1  FuseMode.DISPATCH_FFN_COMBINE (reference deep_ep.cpp:1254-1287; buffer.py:854-869).  Differences from fused_deep_moe that
// matter to a caller: weights arrive in their plain logical shapes [L, H, 2I] / [L, I, H] with gate = columns [0, I) and
// up = [I, 2I) (no fusion-tile permutation; activate_left swish, tests/python/deepep/test_dispatch_ffn_combine.py:168-178),
// scales are fp32 bit patterns widened to int64 (:60-69), `max_output_size` bounds the rows a rank can receive, and the
// second output is the per-local-expert row count [L].  Here the weights are re-laid-out once (cached) into the fusion-tile,
// K-contiguous form the grouped GEMM consumes, then the same launch chain runs.
std::vector<at::Tensor> Buffer::dispatch_ffn_combine(const at::Tensor &x, const at::Tensor &expert_ids, const at::Tensor &weight1,
                                                     const at::Tensor &scale1, const at::Tensor &weight2, const at::Tensor &scale2,
                                                     const std::optional<at::Tensor> &expert_scales, int64_t max_output_size,
                                                     int64_t num_experts, int64_t quant_mode)
{
    require_available();
    EP_HOST_ASSERT_S(shared_expert_rank_num == 0, "dispatch_ffn_combine does not take shared-expert ranks (the reference passes none either, ",
                     "deep_ep.cpp:1254-1287); use fused_deep_moe or the low-latency ops");
    EP_HOST_ASSERT(max_output_size > 0);
    EP_HOST_ASSERT_S(weight1.scalar_type() == at::kChar, "BF16 mode not yet supported for dispatch_ffn_combine");
    fused_common_checks(x, expert_ids, weight1, weight2, quant_mode, expert_scales, "dispatch_ffn_combine");
    const int W = (int)num_ranks, L = (int)num_experts / W, H = (int)x.size(1), T = (int)x.size(0), K = (int)expert_ids.size(1);
    EP_HOST_ASSERT(weight1.size(0) == L and weight2.size(0) == L);
    EP_HOST_ASSERT_S(weight1.size(1) == H and weight2.size(2) == H and weight2.size(1) * 2 == weight1.size(2),
                     "dispatch_ffn_combine expects weight1 [L, hidden, 2*inter] and weight2 [L, inter, hidden]");
    const int64_t N1 = weight1.size(2), I = N1 / 2;
    auto fusion_perm = [&]() {                   // fused row p -> logical output channel: 64 gate channels then their 64 up channels
        at::Tensor p = at::arange(N1, at::dtype(at::kLong).device(x.device()));
        at::Tensor within = p.remainder(128), blk = p.div(128, "floor");
        return at::where(within < 64, blk * 64 + within, blk * 64 + within - 64 + I);
    };
    at::Tensor w1 = prepared_weight(weight1, 2, [&] { return weight1.transpose(1, 2).index_select(1, fusion_perm()).contiguous(); });
    at::Tensor w2 = prepared_weight(weight2, 1, [&] { return weight2.transpose(1, 2).contiguous(); });
    auto as_f32 = [](const at::Tensor &s) {
        if (s.scalar_type() == at::kLong) return s.to(at::kInt).contiguous().view(at::kFloat);   // fp32 bits carried in int64
        return s.to(at::kFloat).contiguous();
    };
    // the re-laid-out scales are layer constants like the weights: prepared once and kept
    at::Tensor s1 = prepared_weight(scale1, 3, [&] { return as_f32(scale1).reshape({L, N1}).index_select(1, fusion_perm()).contiguous(); });
    at::Tensor s2 = prepared_weight(scale2, 4, [&] { return as_f32(scale2).reshape({L, H}).contiguous(); });
    at::Tensor topk_weights = expert_scales->to(at::kFloat).contiguous();
    // every rank may send at most max(T over ranks) tokens; the caller's bound is rows received = max_bs * W * K
    int64_t per_rank = (max_output_size + (int64_t)W * K - 1) / ((int64_t)W * K);
    EP_HOST_ASSERT_S(T <= per_rank, "dispatch_ffn_combine: max_output_size (", max_output_size, ") < tokens * num_ranks * topk");
    auto r = fused_core(x, expert_ids, w1, s1, w2, s2, topk_weights, per_rank, num_experts);
    at::Tensor ends = r[1].reshape({L, W}).select(1, W - 1);                     // inclusive cumulative rows at each expert's end
    at::Tensor counts = ends - at::cat({at::zeros({1}, ends.options()), ends.slice(0, 0, L - 1)});
    return {r[0], counts.to(expert_ids.scalar_type())};
}

// The reference's stage profiler (device timestamp ring + host exporter -> chrome trace_view.json, Ascend950 fused op
// only: deep_ep.cpp:1237-1252, profiling/core/profile_exporter.cpp:421-427) becomes HIP event pairs around every
// kernel of the dispatch/combine chains, on the caller's stream.  Calls = dispatch/combine API calls; the first
// `skip` are ignored, the next `active` are recorded.  end_profile() drains the events, fills get_profile_summary()
// and, if a directory was given, writes <dir>/trace_view_rank<r>.json (chrome://tracing "X" events).
// DEEPEP_ROCTX=1: every kernel chain of dispatch / combine / fused_deep_moe is also a roctx range (host-side push / pop around
// its launches; `rocprofv3 --marker-trace --kernel-trace` then groups the kernels by the same names get_profile_summary() uses).
// libroctx64 is looked up at first use, so there is no link-time dependency and no cost when the switch is off.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *on = getenv("DEEPEP_ROCTX");
        if (!on || !atoi(on)) return;
        for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {      // rocprofv3 listens to the SDK library
            if (void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr, pop = nullptr;
            }
        }
    }
};
const Roctx &roctx()
{
    static const Roctx r;
    return r;
}
}  // namespace

ProfScope::ProfScope(Buffer *b, const char *name, hipStream_t st) : b(b), st(st)
{
    if (roctx().push) roctx().push(name), marked = true;
    if (!b->profile_now()) return;
    Buffer::ProfRec r{name, nullptr, nullptr};
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    hipEventRecord(r.a, st);
    idx = b->profile_recs.size();
    b->profile_recs.push_back(r);
}
ProfScope::~ProfScope()
{
    if (idx != (size_t)-1) hipEventRecord(b->profile_recs[idx].b, st);
    if (marked) roctx().pop();
}

void Buffer::begin_profile(int64_t skip, int64_t active, const std::string &dir)
{
    profile_skip = skip, profile_active = active, profile_calls = 0, profiling = true, profile_dir = dir;
    profile_recs.clear();
    profile_summary.clear();
}

void Buffer::end_profile()
{
    profiling = false;
    if (profile_recs.empty()) return;
    HIP_CHECK(hipEventSynchronize(profile_recs.back().b));
    std::vector<std::tuple<std::string, int64_t, double>> sum;
    std::string trace = "[";
    const hipEvent_t origin = profile_recs.front().a;
    for (auto &r : profile_recs) {
        float ms = 0.f, t0 = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
        if (hipEventElapsedTime(&t0, origin, r.a) != hipSuccess) t0 = 0.f;
        bool found = false;
        for (auto &s : sum)
            if (std::get<0>(s) == r.name) {
                std::get<1>(s) += 1;
                std::get<2>(s) += ms;
                found = true;
            }
        if (!found) sum.emplace_back(r.name, 1, (double)ms);
        if (trace.size() > 1) trace += ",";
        trace += ep_concat("{\"name\":\"", r.name, "\",\"ph\":\"X\",\"pid\":", rank, ",\"tid\":0,\"ts\":", t0 * 1000.0,
                           ",\"dur\":", ms * 1000.0, "}");
    }
    for (auto &r : profile_recs) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    trace += "]";
    profile_recs.clear();
    profile_summary = sum;
    if (!profile_dir.empty()) {
        const std::string path = profile_dir + "/trace_view_rank" + std::to_string(rank) + ".json";
        if (FILE *f = std::fopen(path.c_str(), "w")) {
            std::fwrite(trace.data(), 1, trace.size(), f);
            std::fclose(f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernel-level entry points for the `alltoall` strategies
// ------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor> Buffer::a2a_dispatch_stage(const at::Tensor &x, const at::Tensor &topk_idx,
                                                              int64_t num_experts, const std::string &quant_type)
{
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous() and x.scalar_type() == at::kBFloat16);
    const int T = (int)x.size(0), H = (int)x.size(1), K = (int)topk_idx.size(1), E = (int)num_experts;
    const int qm = quant_mode_of(quant_type != "bf16", quant_type);
    const Layout &lay = layout_for(topk_idx, E);
    const size_t rb = mi_ep_dispatch_row_bytes(H, qm);
    auto rows = at::empty({std::max<int64_t>((int64_t)T * K, 1), (int64_t)rb}, at::dtype(at::kByte).device(x.device()));
    MI_EP_CHECK(mi_ep_dispatch_stage(x.data_ptr(), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt,
                                     lay.send_token_idx_small.data_ptr<int>(), lay.send_data_offset.data_ptr<int>(), T, K, H,
                                     E, (int)rank, qm, rows.data_ptr(), cur_stream()));
    auto cnt_vec = at::empty({E + 1}, at::dtype(at::kInt).device(x.device()));
    cnt_vec.narrow(0, 0, E).copy_(lay.num_tokens_per_expert);
    cnt_vec.narrow(0, E, 1).fill_(T);
    return {rows, cnt_vec};
}

std::tuple<at::Tensor, at::Tensor, std::vector<int64_t>, std::vector<int64_t>, std::vector<int>, int64_t, int64_t>
Buffer::a2a_dispatch_tables(const at::Tensor &cnt_matrix)
{
    EP_HOST_ASSERT(cnt_matrix.dim() == 2 and cnt_matrix.is_contiguous() and cnt_matrix.scalar_type() == at::kInt);
    const int W = (int)num_ranks, E = (int)cnt_matrix.size(1) - 1, L = E / W;
    EP_HOST_ASSERT(cnt_matrix.size(0) == W and E % W == 0);
    auto i32 = at::dtype(at::kInt).device(cnt_matrix.device());
    auto recv_count = at::empty({E}, i32), recv_offset = at::empty({E}, i32), per_e = at::empty({L}, i32);
    auto ego = at::empty({L}, i32), sie = at::empty({E}, i32), ris = at::empty({E}, i32), total = at::empty({1}, i32);
    auto max_bs = at::empty({1}, i32), pull_offset = at::empty({E}, i32);
    MI_EP_CHECK(mi_ep_notify_tables(cnt_matrix.data_ptr<int>(), W, E, (int)rank, 1, recv_count.data_ptr<int>(),
                                    recv_offset.data_ptr<int>(), per_e.data_ptr<int>(), ego.data_ptr<int>(),
                                    sie.data_ptr<int>(), ris.data_ptr<int>(), total.data_ptr<int>(), max_bs.data_ptr<int>(),
                                    pull_offset.data_ptr<int>(), nullptr, cur_stream()));
    auto host = cnt_matrix.to(at::kCPU);          // the one host sync of this transport
    const int32_t *c = host.data_ptr<int32_t>();
    std::vector<int64_t> send_rows((size_t)W, 0), recv_rows((size_t)W, 0);
    std::vector<int> per_expert((size_t)L, 0);
    int64_t tot = 0, mb = 0;
    for (int r = 0; r < W; ++r) {
        for (int le = 0; le < L; ++le) {
            send_rows[(size_t)r] += c[(size_t)rank * (E + 1) + r * L + le];
            const int v = c[(size_t)r * (E + 1) + (int)rank * L + le];
            recv_rows[(size_t)r] += v;
            per_expert[(size_t)le] += v;
            tot += v;
        }
        mb = std::max<int64_t>(mb, c[(size_t)r * (E + 1) + E]);
    }
    real_max_bs = mb;
    return {recv_count, pull_offset, send_rows, recv_rows, per_expert, tot, mb};
}

std::tuple<at::Tensor, std::optional<at::Tensor>, at::Tensor>
Buffer::a2a_dispatch_unpack(const at::Tensor &staging, const std::vector<int64_t> &recv_rows_per_rank,
                            const at::Tensor &recv_count, const at::Tensor &pull_offset, int64_t hidden, int64_t total_recv,
                            const std::string &quant_type, int64_t min_rows, int64_t src_idx_len)
{
    const int W = (int)num_ranks, E = (int)recv_count.size(0), L = E / W, H = (int)hidden;
    const bool use_quant = quant_type != "bf16";
    const int qm = quant_mode_of(use_quant, quant_type);
    const size_t rb = mi_ep_dispatch_row_bytes(H, qm);
    EP_HOST_ASSERT((int64_t)recv_rows_per_rank.size() == W);
    std::vector<const void *> bases((size_t)W);
    int64_t off = 0;
    for (int r = 0; r < W; ++r) {
        bases[(size_t)r] = (const uint8_t *)staging.data_ptr() + (size_t)off * rb;
        off += recv_rows_per_rank[(size_t)r];
    }
    EP_HOST_ASSERT(off == total_recv);
    const int64_t rows = std::max<int64_t>(total_recv == 0 ? 1 : total_recv, min_rows);
    auto dev = staging.device();
    at::Tensor recv_x = use_quant ? at::empty({rows, H}, at::dtype(payload_dtype(qm)).device(dev))
                                  : at::empty({rows, H}, at::dtype(at::kBFloat16).device(dev));
    at::Tensor scales = at::empty({rows}, at::dtype(at::kFloat).device(dev));
    at::Tensor src_idx = at::empty({std::max<int64_t>(rows * 3, src_idx_len)}, at::dtype(at::kInt).device(dev));
    MI_EP_CHECK(mi_ep_dispatch_pull(bases.data(), recv_count.data_ptr<int>(), pull_offset.data_ptr<int>(), W, L, H, qm,
                                    (int)total_recv, recv_x.data_ptr(), use_quant ? scales.data_ptr<float>() : nullptr,
                                    src_idx.data_ptr<int>(), cur_stream()));
    return {recv_x, use_quant ? std::optional<at::Tensor>(scales) : std::nullopt, src_idx};
}

std::tuple<at::Tensor, std::vector<int64_t>> Buffer::a2a_combine_pack(const at::Tensor &x, const at::Tensor &send_head)
{
    EP_HOST_ASSERT(x.dim() == 2 and x.is_contiguous() and x.scalar_type() == at::kBFloat16);
    const int W = (int)num_ranks, E = (int)send_head.size(0), L = E / W, H = (int)x.size(1);
    auto packed = at::empty_like(x);
    auto rows_per_src = at::empty({W}, at::dtype(at::kInt).device(x.device()));
    MI_EP_CHECK(mi_ep_combine_pack(x.data_ptr(), send_head.data_ptr<int>(), W, L, H, (int)x.size(0), packed.data_ptr(),
                                   rows_per_src.data_ptr<int>(), cur_stream()));
    auto host = rows_per_src.to(at::kCPU);
    std::vector<int64_t> v((size_t)W);
    for (int r = 0; r < W; ++r) v[(size_t)r] = host.data_ptr<int32_t>()[r];
    return {packed, v};
}

std::tuple<at::Tensor, at::Tensor, std::vector<int64_t>> Buffer::a2a_combine_prepare(const at::Tensor &topk_idx,
                                                                                     int64_t num_experts)
{
    const Layout &lay = layout_for(topk_idx, (int)num_experts);
    const int W = (int)num_ranks, L = (int)num_experts / W;
    auto host = lay.num_tokens_per_expert.to(at::kCPU);
    std::vector<int64_t> v((size_t)W, 0);
    for (int r = 0; r < W; ++r)
        for (int le = 0; le < L; ++le) v[(size_t)r] += host.data_ptr<int32_t>()[r * L + le];
    return {lay.send_data_offset, lay.send_token_idx_small, v};
}

at::Tensor Buffer::a2a_combine_reduce(const at::Tensor &returned_rows, const at::Tensor &topk_idx,
                                      const std::optional<at::Tensor> &topk_weights, const at::Tensor &send_data_offset,
                                      const at::Tensor &send_token_idx_small, int64_t hidden, int64_t num_experts)
{
    const int T = (int)topk_idx.size(0), K = (int)topk_idx.size(1), H = (int)hidden;
    auto out = at::empty({T, H}, at::dtype(at::kBFloat16).device(topk_idx.device()));
    MI_EP_CHECK(mi_ep_combine_reduce(returned_rows.data_ptr(), topk_idx.data_ptr(), topk_idx.scalar_type() == at::kInt,
                                     topk_weights.has_value() ? topk_weights->data_ptr<float>() : nullptr,
                                     send_data_offset.data_ptr<int>(), send_token_idx_small.data_ptr<int>(), T, K, H,
                                     (int)num_experts, out.data_ptr(), nullptr, 0, nullptr, nullptr, 0, 0, 1, cur_stream()));
    return out;
}

}  // namespace deep_ep
