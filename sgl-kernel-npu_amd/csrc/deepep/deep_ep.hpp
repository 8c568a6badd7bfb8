// MI355X host runtime behind `deep_ep.Buffer` (pybind module deep_ep_cpp).
// Drop-in for the reference's deep_ep::Buffer (csrc/deepep/deep_ep.hpp:18-143, pybind_extension.cpp:31-55): same
// constructor, same method names, same positional argument orders and return tuples.  Device work is the HIP kernels
// of include/mi_ep.h; the HCCL window is replaced by a hipIpc-mapped symmetric window over xGMI.
#pragma once
#include <array>
#include <functional>
#include <map>
#include <optional>
#include <string>
#include <tuple>
#include <vector>

#include <ATen/ATen.h>
#include <hip/hip_runtime_api.h>

#include "config.hpp"
#include "exception.hpp"

namespace deep_ep {

class Buffer {
public:
    Buffer(int64_t rank, int64_t num_ranks, int64_t num_nvl_bytes, int64_t num_rdma_bytes, bool low_latency_mode,
           std::string moe_all_to_all_group_name);
    ~Buffer() noexcept(false);

    // ---- MI355X bootstrap of the symmetric window (the reference gets its window from HCCL by group name,
    // buffer.py:67-83; here Python all-gathers these handles over the ProcessGroup, like upstream DeepEP's
    // get_local_ipc_handle()/sync()).
    int get_local_device_id() const { return device_id; }
    // PCI bus id of this rank's GPU ("0000:05:00.0"): the bootstrap compares them -- a device index says nothing under per-process
    // HIP_VISIBLE_DEVICES.  Ranks that SHARE a GPU (the one-GPU test / dry-run setups) keep the low-latency calls on their three-launch forms:
    // the consuming launches of the two-launch forms wait for rows inside many workgroups, which is fine when every rank owns its GPU (the
    // producing launches run elsewhere) and can starve the producers of the other ranks when they all queue on one.
    std::string get_local_device_bus_id() const;
    void set_ranks_share_device(bool shared) { ranks_share_device = shared; }
    bool get_ranks_share_device() const { return ranks_share_device; }
    // The window is kNumSegs allocations (control area + one per family, each below the 2 GiB that hipIpcOpenMemHandle can map).
    std::string get_local_ipc_handle() const;         // kNumSegs x hipIpcMemHandle_t bytes
    std::vector<int64_t> get_local_window_ptrs() const;      // the kNumSegs segment bases
    int64_t get_window_bytes() const { return window_bytes; }
    // handles[r]: ipc handle bytes of rank r (ignored for r == rank or when local_ptrs[r] is not empty);
    // local_ptrs[r]: segment bases of rank r when it lives in this process, else empty.
    void sync(const std::vector<std::string> &handles, const std::vector<std::vector<int64_t>> &local_ptrs);

    // MI355X: transport of normal-mode dispatch, "push" (remote writes into the receivers' windows) or "pull" (receivers read
    // the senders' windows).  Every rank of the group must select the same one before its next dispatch.
    void set_dispatch_transport(const std::string &name);
    void set_local_row_paths(bool dispatch_local, bool combine_local);
    std::vector<bool> get_local_row_paths() const { return {dispatch_local_rows, combine_local_rows_enabled}; }
    // MI355X only: fused_deep_moe / dispatch_ffn_combine at prefill sizes multiply the staged token rows in place (a row-offset table) instead
    // of a K-fold gathered copy; default from MI_EP_FUSED_GATHER (0 = gather).  Results are identical either way.
    // MI355X only: GEMM1 of fused_deep_moe / dispatch_ffn_combine requantises its rows in its epilogue (mi_ep_moe_gemm1_swiglu_quant) instead
    // of writing fp32 rows for a rowquant launch.  Results are identical either way.  OPT-IN (MI_EP_FUSED_REQUANT=1): at BASELINE C5 it measured
    // 1.586-1.591 ms against 1.575-1.583 for the two launches on the same box -- the 16 column-tile workgroups of a row block then run in
    // lockstep, and the sum over tiles of the slowest of 16 costs what the rowquant launch and its fp32 round trip cost.
    void set_fused_requant(bool on) { fused_requant = on; }
    bool get_fused_requant() const { return fused_requant; }
    int get_gemm_xcds() const { return gemm_xcds; }
    void set_fused_rows_in_place(bool on) { fused_rows_in_place = on; }
    bool get_fused_rows_in_place() const { return fused_rows_in_place; }
    std::string get_dispatch_transport() const { return dispatch_transport == kTransportPush ? "push" : "pull"; }
    bool self_test(int64_t test_timeout_ms);     // collective: every rank calls it after sync()
    // Second leg (collective, after self_test passed everywhere): the in-launch hand-off the two-launch low-latency forms rest on
    // (mi_ep_selftest_inlaunch: tag / flag word behind a drained write-through payload, polled and read inside one launch, four rounds over
    // both ping-pong halves).  skip_payload_from_round >= 0 is the test hook of the same name.  A failure anywhere makes deep_ep.Buffer call
    // set_two_launch_forms(false) on every rank: the low-latency calls then keep their three-launch forms (kernel boundaries carry the
    // ordering), not the alltoall strategies.
    bool self_test_in_launch(int64_t test_timeout_ms, int64_t skip_payload_from_round);
    void set_two_launch_forms(bool ok) { two_launch_forms_ok = ok; }
    bool get_two_launch_forms() const { return two_launch_forms_ok; }
    // the forms of a dispatch and of a combine behind a dispatch (0 three launches, 1 tail-fused, 2 two launches) ...
    std::vector<int64_t> get_low_latency_default_forms() const
    {
        return {ll_launch_form("MI_EP_LL_FUSED_COUNTS"), ll_launch_form("MI_EP_COMBINE_FUSED")};
    }
    // ... and the forms the NEXT low_latency_dispatch / low_latency_combine would take (a combine behind a combine: three launches)
    std::vector<int64_t> get_low_latency_launch_forms() const
    {
        return {ll_launch_form("MI_EP_LL_FUSED_COUNTS"), last_ll_call_was_combine ? 0 : ll_launch_form("MI_EP_COMBINE_FUSED")};
    }
    bool is_available() const { return available; }
    // false only when DEEPEP_WINDOW_FINEGRAINED=0 forced a coarse-grained window: peers' stores are then not guaranteed to
    // become visible inside a running kernel, so deep_ep.Buffer selects the alltoall (RCCL) strategies for W > 1.
    bool is_window_fine_grained() const { return window_fine_grained; }
    int get_num_rdma_ranks() const { return 1; }
    int get_rdma_rank() const { return 0; }

    std::tuple<at::Tensor, std::optional<at::Tensor>, at::Tensor, at::Tensor, std::optional<EventHandle>>
    get_dispatch_layout(const at::Tensor &topk_idx, int num_experts, std::optional<EventHandle> &previous_event,
                        bool async, bool allocate_on_comm_stream);
    at::Tensor get_notify_send_data();
    void clean_low_latency_buffer(int num_max_dispatch_tokens_per_rank, int hidden, int num_experts);

    std::tuple<at::Tensor, std::optional<at::Tensor>, std::optional<at::Tensor>, std::optional<at::Tensor>,
               std::vector<int>, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, std::optional<EventHandle>>
    intranode_dispatch(const at::Tensor &x, const std::optional<at::Tensor> &x_scales,
                       const std::optional<at::Tensor> &topk_idx, const std::optional<at::Tensor> &topk_weights,
                       const std::optional<at::Tensor> &num_tokens_per_rank, const at::Tensor &is_token_in_rank,
                       const std::optional<at::Tensor> &num_tokens_per_expert, int cached_num_recv_tokens,
                       const std::optional<at::Tensor> &cached_rank_prefix_matrix,
                       const std::optional<at::Tensor> &cached_channel_prefix_matrix,
                       const std::optional<at::Tensor> &dispatch_wait_recv_cost_stats, int expert_alignment,
                       int num_worst_tokens, const Config &config, std::optional<EventHandle> &previous_event, bool async,
                       bool allocate_on_comm_stream, bool use_quant, const std::string &quant_type);

    std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor, at::Tensor>
    notify_verify(const at::Tensor &x, const std::optional<at::Tensor> &x_scales, const std::optional<at::Tensor> &topk_idx,
                  const std::optional<at::Tensor> &topk_weights, const std::optional<at::Tensor> &num_tokens_per_rank,
                  const at::Tensor &is_token_in_rank, const std::optional<at::Tensor> &num_tokens_per_expert,
                  int cached_num_recv_tokens, const std::optional<at::Tensor> &cached_rank_prefix_matrix,
                  const std::optional<at::Tensor> &cached_channel_prefix_matrix,
                  const std::optional<at::Tensor> &dispatch_wait_recv_cost_stats, int expert_alignment,
                  int num_worst_tokens, const Config &config, std::optional<EventHandle> &previous_event, bool async,
                  bool allocate_on_comm_stream, bool use_quant);

    std::tuple<at::Tensor, std::optional<at::Tensor>, std::optional<EventHandle>>
    intranode_combine(const at::Tensor &x, const at::Tensor &topk_idx, const std::optional<at::Tensor> &topk_weights,
                      const at::Tensor &src_idx, const at::Tensor &send_head,
                      const std::optional<at::Tensor> &combine_send_cost_stats);

    std::tuple<at::Tensor, std::optional<at::Tensor>, at::Tensor, at::Tensor, at::Tensor, std::optional<EventHandle>,
               std::optional<std::function<void()>>>
    low_latency_dispatch(const at::Tensor &x, const at::Tensor &topk_idx,
                         const std::optional<at::Tensor> &cumulative_local_expert_recv_stats,
                         int64_t num_max_dispatch_tokens_per_rank, int64_t num_experts, bool use_fp8, bool round_scale,
                         bool use_ue8m0, bool use_mxfp4, bool async, bool return_recv_hook,
                         const std::string &quant_mode_name);

    std::tuple<at::Tensor, std::optional<EventHandle>, std::optional<std::function<void()>>>
    low_latency_combine(const at::Tensor &x, const at::Tensor &topk_idx, const at::Tensor &topk_weights,
                        const at::Tensor &src_info, const at::Tensor &layout_range,
                        int64_t num_max_dispatch_tokens_per_rank, int64_t num_experts,
                        const at::Tensor &packed_recv_count, bool zero_copy, bool async, bool return_recv_hook,
                        const std::optional<at::Tensor> &out);

    // Multi-node entry points of the reference (Ascend910B layered HCCS+RDMA): one 8-GPU xGMI node has a single
    // "rdma rank", so these are never reached through deep_ep.Buffer; calling them directly is an error.
    void internode_unsupported() const;

    std::vector<at::Tensor> fused_deep_moe(const at::Tensor &x, const at::Tensor &expert_ids,
                                           const at::Tensor &gmm1_permuted_weight,
                                           const at::Tensor &gmm1_permuted_weight_scale, const at::Tensor &gmm2_weight,
                                           const at::Tensor &gmm2_weight_scale,
                                           const std::optional<at::Tensor> &expert_scales_optional,
                                           int64_t num_max_dispatch_tokens_per_rank, int64_t num_experts,
                                           int64_t quant_mode, bool profile_enable);
    std::vector<at::Tensor> dispatch_ffn_combine(const at::Tensor &x, const at::Tensor &expert_ids,
                                                 const at::Tensor &gmm1_weight, const at::Tensor &gmm1_scale,
                                                 const at::Tensor &gmm2_weight, const at::Tensor &gmm2_scale,
                                                 const std::optional<at::Tensor> &expert_scales, int64_t max_output_size,
                                                 int64_t num_experts, int64_t quant_mode);
    void begin_profile(int64_t num_profile_skip_launches, int64_t num_profile_active_launches,
                       const std::string &profile_trace_dir);
    void end_profile();
    // MI355X: per-kernel device time (HIP events on the caller's stream) gathered between begin_profile/end_profile:
    // name -> (launch count, total milliseconds).  Valid after end_profile().
    std::vector<std::tuple<std::string, int64_t, double>> get_profile_summary() const { return profile_summary; }
    // MI355X only: shader clock (GHz) and duration (us) of the first workgroup of the LAST grouped-GEMM launches {gemm1, gemm2, gemm2 + push}
    std::vector<std::pair<double, double>> get_gemm_clock() const;

    // ---- kernel-level entry points used by the `alltoall` strategies (torch.distributed / RCCL moves the bytes,
    // these pack and unpack them).  Mirrors what the reference's AlltoAll strategies get from torch_npu routing ops
    // (strategies/normal_strategy.py:481-790).
    std::tuple<at::Tensor, at::Tensor> a2a_dispatch_stage(const at::Tensor &x, const at::Tensor &topk_idx,
                                                          int64_t num_experts, const std::string &quant_type);
    // cnt_matrix [W, E+1] int32 on device -> (recv_count, pull_offset, send_rows_per_rank, recv_rows_per_rank,
    //                                        recv_tokens_per_expert(list), total_recv, max_bs)
    std::tuple<at::Tensor, at::Tensor, std::vector<int64_t>, std::vector<int64_t>, std::vector<int>, int64_t, int64_t>
    a2a_dispatch_tables(const at::Tensor &cnt_matrix);
    std::tuple<at::Tensor, std::optional<at::Tensor>, at::Tensor>
    a2a_dispatch_unpack(const at::Tensor &staging, const std::vector<int64_t> &recv_rows_per_rank,
                        const at::Tensor &recv_count, const at::Tensor &pull_offset, int64_t hidden, int64_t total_recv,
                        const std::string &quant_type, int64_t min_rows, int64_t src_idx_len);
    std::tuple<at::Tensor, std::vector<int64_t>> a2a_combine_pack(const at::Tensor &x, const at::Tensor &send_head);
    // -> (send_data_offset, send_token_idx_small, rows_sent_to_rank)
    std::tuple<at::Tensor, at::Tensor, std::vector<int64_t>> a2a_combine_prepare(const at::Tensor &topk_idx,
                                                                                 int64_t num_experts);
    at::Tensor a2a_combine_reduce(const at::Tensor &returned_rows, const at::Tensor &topk_idx,
                                  const std::optional<at::Tensor> &topk_weights, const at::Tensor &send_data_offset,
                                  const at::Tensor &send_token_idx_small, int64_t hidden, int64_t num_experts);

private:
    struct Layout {
        at::Tensor num_tokens_per_rank, num_tokens_per_expert, is_token_in_rank, send_token_idx_small, send_data_offset,
            workspace;
        int64_t T = -1, K = -1, E = -1;
        // The tensor the layout was computed for is HELD (its storage cannot be freed and recycled for another tensor while
        // the stash is alive) together with its version counter (in-place rewrites invalidate the stash).
        at::Tensor idx;
        int64_t idx_version = -1;
        // Inference tensors (torch.inference_mode(), the reference's own test harness runs every rank under it,
        // tests/python/deepep/test_fused_deep_moe_a5.py:723) carry no version counter: at::Tensor::_version() throws for
        // them.  Their version is recorded as -1 and a stash with version -1 never matches (an in-place rewrite could
        // not be seen), so such callers always get a freshly computed layout.
        static int64_t version_of(const at::Tensor &t) { return t.is_inference() ? -1 : (int64_t)t._version(); }
        bool matches(const at::Tensor &t, int64_t num_experts) const
        {
            return idx.defined() && idx_version >= 0 && idx.data_ptr() == t.data_ptr() && idx.scalar_type() == t.scalar_type() &&
                   idx_version == version_of(t) && T == t.size(0) && K == t.size(1) && E == num_experts;
        }
    };
    Layout run_layout(const at::Tensor &topk_idx, int num_experts);
    uint32_t *layout_sync_words(const at::Device &dev);
    uint64_t layout_calls = 0;
    const Layout &layout_for(const at::Tensor &topk_idx, int num_experts);

    void check_status(const char *where);
    int64_t wait_summary(const char *where);     // host spin on the pinned summary word written by notify_tables
    int32_t summary_word(int i, const char *where);  // one further word of it (the kernel orders nothing among them)
    uint8_t *family_base(int family) const;
    uint64_t *epoch_ctr(int family) const;
    std::vector<void *> peer_ptrs(size_t offset) const;                 // into the control segment
    std::vector<void *> peer_family_bases(int family) const;
    void require_available() const;

    int64_t rank, num_ranks, num_nvl_bytes, num_rdma_bytes;
    bool low_latency_mode;
    std::string group_name;
    int device_id = 0;
    bool available = false;
    int timeout_ms = 30000;

    // symmetric window
    uint8_t *window = nullptr;
    int64_t window_bytes = 0;
    bool window_fine_grained = false;
    enum { kSegCtrl = 0, kNumSegs = 4 };   // control area, then one segment per family (dispatch, combine, low-latency dispatch)
    std::array<uint8_t *, kNumSegs> seg_base{};        // own segments (window == seg_base[kSegCtrl])
    std::array<size_t, kNumSegs> seg_bytes{};
    std::vector<std::array<uint8_t *, kNumSegs>> peer_seg;   // [W] mapped segment bases of every rank (own = seg_base)
    std::vector<bool> peer_opened;         // true when mapped through hipIpcOpenMemHandle
    size_t region_bytes = 0;               // each of the 6 data regions
    // pinned host words the kernels write with system scope
    int32_t *summary_host = nullptr;       // [2 + L] see mi_ep_notify_tables
    int32_t *status_host = nullptr;        // [4]
    int32_t *summary_dev = nullptr, *status_dev = nullptr;

    enum { kTransportPull = 0, kTransportPush = 1 };
    int dispatch_transport = kTransportPull;
    uint64_t selftest_epoch = 0;           // (dispatch / combine / low-latency call counters live on the device: epoch_ctr())
    Layout stash;                          // hidden state coupling of the reference (deep_ep.cpp:170-172,321)
    int64_t real_max_bs = 0;
    // MOE_SHARED_EXPERT_RANK_NUM (reference deep_ep.cpp:62): the first S ranks hold the shared expert; low-latency ops and fused_deep_moe only
    int64_t shared_expert_rank_num = 0;
    // the (K+1)-selection view of a routing table under shared-expert ranks (mi_ep_shared_expert_map); E / K / L of the renamed experts
    struct SharedView {
        at::Tensor idx, weights;
        int E = 0, K = 0, L = 0, local_experts = 0;
    };
    SharedView shared_view(const at::Tensor &topk_idx, const float *weights, bool want_weights, int64_t num_experts, hipStream_t st) const;
    int64_t profile_skip = 0, profile_active = 0, profile_calls = 0;
    bool profiling = false;
    std::string profile_dir;
    int64_t last_recv_rows = 0;        // rows received by the previous host-synchronised dispatch (sizes the speculative pull)
    struct NotifyTables {
        at::Tensor cnt, recv_count, recv_offset, recv_tokens_per_expert, expert_global_offset, srcrank_in_expert_offset,
            r_in_srcrank_offset, total_recv_token, max_bs, pull_offset;
    };
    NotifyTables alloc_notify_tables(int W, int E, int L, const at::TensorOptions &i32);
    struct DispatchExchange {              // device half of one normal-mode dispatch (stage + notify exchange)
        NotifyTables nt;
        std::vector<void *> src_bases;     // where the receive side finds source s's token rows + index (half 0 of the ping-pong)
        size_t slab_bytes = 0;
        bool push = false;
        // what the token-wise gather of this rank's own rows needs (mi_ep_dispatch_pull_local)
        at::Tensor topk_idx, send_token_idx_small, num_tokens_per_expert;
        int num_tokens = 0, num_experts = 0;
    };
    DispatchExchange dispatch_exchange(const at::Tensor &x, const at::Tensor &topk_idx, const Layout &lay, int E, int qm,
                                       bool want_summary, int32_t *wait_stats, hipStream_t st);
    void dispatch_pull(const DispatchExchange &ex, int H, int K, int L, int qm, int64_t rows_alloc, const at::TensorOptions &x_opts,
                       at::Tensor &rx, at::Tensor &rs, at::Tensor &src_idx, hipStream_t st);
    at::Tensor combine_finish(const at::Tensor &topk_idx, const float *topk_weights, int H, int E, const at::TensorOptions &opts,
                              const char *reduce_name, hipStream_t st, const at::Tensor &x_local = at::Tensor(),
                              const at::Tensor &local_row = at::Tensor(), int signalled = 0);
    // push + signal + wait in one launch (mi_ep_combine_push_signal_wait); MI_EP_COMBINE_FUSED=0: the separate signal_wait launch
    void combine_push_rows(const at::Tensor &x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int H, int K,
                           const at::Tensor &local_row, const char *name, hipStream_t st, int &signalled, bool may_flag_rows = false);
    uint32_t *arrive_word();
    uint64_t arrive_calls = 0;
    bool ranks_share_device = false;
    bool two_launch_forms_ok = true;       // cleared when the in-launch self-test leg failed on any rank
    // A two-launch combine is only safe behind a dispatch: its reduce waits for the ranks that serve ITS tokens, not for everybody, so with
    // two combines back to back a rank that waits on nobody could run a call ahead and rewrite a ping-pong half its owner still reduces.
    // The dispatch's count exchange is all-to-all and closes that; a combine that directly follows another combine on this Buffer takes the
    // three-launch form (whose signal / wait is all-to-all).  Every rank makes the same calls, so every rank switches together.
    bool last_ll_call_was_combine = false;
    uint64_t last_ll_dispatch_capture = 0;      // graph capture the last low_latency_dispatch was recorded into (0: it ran eagerly)
    // launch form of the low-latency dispatch / combine: the env value if set, else 2 (two launches, nothing between them) when every rank owns
    // its GPU and 0 (three launches) when ranks share one
    int ll_launch_form(const char *env_name) const;
    at::Tensor combine_local_rows(const at::Tensor &topk_idx) const;
    struct LocalRowEntry {
        at::Tensor src_idx, rows;
        int T, K;
    };
    std::vector<LocalRowEntry> local_row_stash;
    void remember_local_rows(const at::Tensor &src_idx, const at::Tensor &rows, int T, int K);
    at::Tensor recall_local_rows(const at::Tensor &src_idx, int T, int K) const;
    // defaults from MI_EP_DISPATCH_LOCAL / MI_EP_COMBINE_LOCAL (0 = off), see set_local_row_paths()
    bool dispatch_local_rows = !(getenv("MI_EP_DISPATCH_LOCAL") && atoi(getenv("MI_EP_DISPATCH_LOCAL")) == 0);
    bool combine_local_rows_enabled = !(getenv("MI_EP_COMBINE_LOCAL") && atoi(getenv("MI_EP_COMBINE_LOCAL")) == 0);
    bool fused_rows_in_place = !(getenv("MI_EP_FUSED_GATHER") && atoi(getenv("MI_EP_FUSED_GATHER")) == 0);
    bool fused_requant = getenv("MI_EP_FUSED_REQUANT") && atoi(getenv("MI_EP_FUSED_REQUANT")) != 0;
    int gemm_xcds = 1;
    // fused paths: shared launch chain + one-off weight re-layout cache (keyed by storage pointer and kind)
    std::vector<at::Tensor> fused_core(const at::Tensor &x, const at::Tensor &expert_ids, const at::Tensor &w1,
                                       const at::Tensor &s1, const at::Tensor &w2, const at::Tensor &s2,
                                       const at::Tensor &topk_weights, int64_t num_max_dispatch_tokens_per_rank,
                                       int64_t num_experts);
    at::Tensor prepared_weight(const at::Tensor &w, int kind, const std::function<at::Tensor()> &make);
    struct WeightKey {
        const void *ptr;
        int kind;
        bool operator<(const WeightKey &o) const { return ptr != o.ptr ? ptr < o.ptr : kind < o.kind; }
    };
    struct WeightEntry {
        int64_t version, numel;
        at::Tensor src;        // the caller's tensor is held: its address cannot be recycled for a different weight
        at::Tensor t;
    };
    std::map<WeightKey, WeightEntry> weight_cache_;
  public:
    void clear_weight_cache() { weight_cache_.clear(); }   // MI355X only: after an in-place weight update of inference tensors
  private:

    struct ProfRec {
        const char *name;
        hipEvent_t a, b;
    };
    std::vector<ProfRec> profile_recs;
    std::vector<std::tuple<std::string, int64_t, double>> profile_summary;
    bool profile_now() const { return profiling && profile_calls > profile_skip && profile_calls <= profile_skip + profile_active; }
    friend struct ProfScope;
};

// RAII: records a HIP event pair around one kernel launch chain when profiling is active
struct ProfScope {
    Buffer *b;
    size_t idx = (size_t)-1;
    hipStream_t st;
    bool marked = false;
    ProfScope(Buffer *b, const char *name, hipStream_t st);
    ~ProfScope();
};

}  // namespace deep_ep
