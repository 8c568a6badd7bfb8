/*
 * mi_ep.h -- C-ABI of the MI355X (gfx950) expert-parallel dispatch/combine kernels.
 *
 * This is the drop-in boundary that replaces the reference's device-operator layer
 * (sgl-kernel-npu, Ascend `aclnn*` two-phase C API):
 *
 *   mi_ep_dispatch_layout      <- aclnnDispatchLayout        csrc/deepep/ops/op_host/op_api/aclnn_dispatch_layout.h:29-41
 *                                 (kernel csrc/deepep/ops/op_kernel/dispatch_layout.h:81-219)
 *   mi_ep_notify_post/_wait,
 *   mi_ep_notify_tables        <- aclnnNotifyDispatch        csrc/deepep/ops/op_host/op_api/aclnn_notify_dispatch.h:36-52
 *                                 (kernel csrc/deepep/ops/op_kernel/notify_dispatch.h:110-132)
 *   mi_ep_dispatch_stage,
 *   mi_ep_dispatch_pull        <- aclnnCamMoeDispatchNormal  csrc/deepep/ops/op_host/op_api/aclnn_cam_moe_dispatch_normal.h:10-21
 *                                 (kernel csrc/deepep/ops/op_kernel/cam_moe_dispatch_normal.h:764-783)
 *   mi_ep_combine_push,
 *   mi_ep_combine_reduce       <- aclnnCamMoeCombineNormal   csrc/deepep/ops/op_host/op_api/aclnn_cam_moe_combine_normal.h:30-45
 *                                 (kernel csrc/deepep/ops/op_kernel/cam_moe_combine_normal.h:446-452)
 *                                 and aclnnMoeLowLatencyCombineV2 (csrc/deepep/deep_ep.cpp:1080)
 *   mi_ep_ll_dispatch_send,
 *   mi_ep_ll_dispatch_recv     <- aclnnMoeLowLatencyDispatchV2 csrc/deepep/deep_ep.cpp:983
 *                                 (kernel csrc/deepep/ops/op_kernel/moe_distribute_dispatch_v2.h:1477-1490)
 *   mi_ep_signal / mi_ep_wait  <- the window flag protocol   csrc/deepep/ops/op_kernel/cam_moe_dispatch_normal.h:496-502,584-631
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every pointer is a DEVICE pointer
 * unless its name ends in `_host`.  Every call only enqueues work on `stream` (a hipStream_t passed
 * as void*), never allocates, never synchronises and never throws.  Return value: 0 = enqueued,
 * negative = MI_EP_E* argument / launch error (nothing enqueued).
 *
 * Device-resident epochs (graph replay).  Entry points that take `epoch_ctr` (+ a parity stride) can run without any host-side
 * call counter: `epoch_ctr` points at a device word of the rank's own control area that holds the number of COMPLETED calls of
 * the kernel family (normal dispatch / combine / low-latency dispatch).  The call's epoch is *epoch_ctr + 1; every window pointer
 * argument is then the base of ping-pong half 0 and the kernel adds (epoch & 1) * stride itself; the family's single-workgroup
 * exchange kernel (notify_exchange_tables / ll_post_recv / signal_wait) stores the new count when it is done, and the consume
 * side launched after it (pull_indexed / the pull inside ll_post_recv / combine_reduce) reads the counter as is.  Nothing the
 * host passes depends on how many calls ran before, so a captured HIP graph replays (the reference keeps its ping-pong word in the
 * window for the same reason: cam_moe_dispatch_normal.h:273-286, notify_dispatch.h:924-938).  epoch_ctr == NULL: the explicit
 * epoch / pointers are used as given and the stride is ignored.
 *
 * Cross-rank data movement is one-sided through "windows": every rank owns a buffer that all
 * peers can address (hipIpc-mapped over xGMI, or plain pointers when several ranks live in one
 * process).  Normal mode, the fused ops and the three-launch low-latency forms never spin on a peer
 * inside a data-moving launch: hand-offs are post-kernel -> mi_ep_signal -> (peer) mi_ep_wait ->
 * consume-kernel, so kernel boundaries carry the release/acquire and only the 8-byte flag words
 * need system-scope atomics.  The two-launch low-latency forms (mi_ep_ll_dispatch_layout_send_tagged +
 * mi_ep_ll_wait_pack, mi_ep_combine_push_flagged + mi_ep_combine_reduce_flagged) hand rows over inside
 * running launches: write-through payload, drain, tag / flag word; consumer: system-scope poll, then
 * system-scope payload loads; mi_ep_selftest_inlaunch checks exactly that.
 */
#ifndef MI_EP_H_
#define MI_EP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_EP_MAX_RANKS 64
#define MI_EP_MAX_TOPK 16
#define MI_EP_MAX_HIDDEN 8192 /* reference limit, cam_moe_dispatch_normal_tiling.cc:75-87 */
#define MI_EP_ROW_META_BYTES 16 /* {f32 scale, i32 token, i32 k, i32 src_rank} appended to every staged row */
/* distance between staged rows: payload + meta rounded up to whole 128-byte lines, so that every row starts on a line (the grouped GEMM of
 * fused_deep_moe reads staged rows in place, 128 bytes of K per request: rows 7184 bytes apart cost it 5 %, mi_ep_moe_gemm1_swiglu_rows) */
#define MI_EP_ROW_STRIDE(payload_bytes) ((((size_t)(payload_bytes)) + MI_EP_ROW_META_BYTES + 127) & ~(size_t)127)

#define MI_EP_OK 0
#define MI_EP_EINVAL (-1)
#define MI_EP_ELAUNCH (-2)
#define MI_EP_ESIZE (-3)        /* a layout does not fit an addressing limit of the kernel asked for; the caller takes the general path */

/* payload modes of the dispatch kernels */
#define MI_EP_QUANT_NONE 0      /* bf16 rows */
#define MI_EP_QUANT_INT8 1      /* s = 127/(amax + 1e-12)   (normal mode, cam_moe_dispatch_normal.h:326-363) */
#define MI_EP_QUANT_INT8_NOEPS 2 /* s = 127/amax             (low-latency, moe_distribute_dispatch_v2.h:1006-1033) */
#define MI_EP_QUANT_FP8_E4M3 3  /* per-token OCP FP8 E4M3 (quant_mode "pertoken_fp8_e4m3"): s = amax > 0 ? 448/amax : 1, q = e4m3fn(x * s)
                                 * round-to-nearest-even, scale_out = 1/s (moe_distribute_dispatch_v2_a5.h:1109-1157, the reference's
                                 * Ascend950-only mode; gfx950 converts natively: v_cvt_pk_fp8_f32).  Row layout as INT8: H bytes + meta */

/* library / build identification ("gfx950") */
const char *mi_ep_version(void);

/* distance between staged dispatch rows: MI_EP_ROW_STRIDE(hidden * (1 or 2)); the meta words sit right behind the payload */
size_t mi_ep_dispatch_row_bytes(int hidden, int quant_mode);
/* bytes of one combine slot row: hidden * 2 rounded up to 16 */
size_t mi_ep_combine_row_bytes(int hidden);

/* ---- A1 layout ------------------------------------------------------------------------------
 * topk_idx [T,K] int64 (ids < 0 or >= E are "no selection").  Outputs (int32):
 *   num_tokens_per_rank [W], num_tokens_per_expert [E], is_token_in_rank [T,W],
 *   send_token_idx_small [T,K]  (rank of the pair among earlier row-major pairs of the same expert; 0 at invalid ids),
 *   send_data_offset [E]        (exclusive prefix of num_tokens_per_expert; reference notify_dispatch.h:185-198).
 * workspace: mi_ep_dispatch_layout_workspace(T,K,E) bytes, contents irrelevant.  idx_is_i32 != 0: topk_idx is int32.
 * sync_words: NULL, or two uint32 words in device memory that the CALLER zero-initialises once and then only lends to this function
 * -- ONE LAUNCH IN FLIGHT PER PAIR: calls that may overlap (different streams) must be lent different pairs; the host runtime deals
 * them from a ring.  Batches of more than 1024 tokens then run as ONE launch (workgroups of 1024 tokens meeting at a self-resetting
 * grid barrier; at most 128 workgroups, which this 256-CU part keeps co-resident) instead of three; NULL keeps the three launches.
 * Results are identical.
 * status: NULL, or a device-visible word: if the grid barrier is not passed within 2 s (a workgroup that never became resident) or the
 * arrival count is seen above the grid size (a pair of sync words lent twice, or not zero when lent) the kernel stores MI_EP_STATUS_LAYOUT_BARRIER there, sets the three count tables to -1 and returns;
 * it never hangs and never returns plausible-looking garbage silently. */
#define MI_EP_STATUS_LAYOUT_BARRIER 6000
size_t mi_ep_dispatch_layout_workspace(int num_tokens, int num_topk, int num_experts);
int mi_ep_dispatch_layout(const void *topk_idx, int idx_is_i32, int num_tokens, int num_topk, int num_experts,
                          int num_ranks, int32_t *num_tokens_per_rank, int32_t *num_tokens_per_expert,
                          int32_t *is_token_in_rank, int32_t *send_token_idx_small, int32_t *send_data_offset,
                          void *workspace, size_t workspace_bytes, uint32_t *sync_words, int32_t *status, void *stream);

/* ---- flags ----------------------------------------------------------------------------------
 * Each rank owns `uint64_t flags[nslots]`; slot s of rank d is written only by rank s.
 * mi_ep_signal: for every d < W store `epoch` into peer_flags[d][my_rank] (system-scope release).
 * mi_ep_wait:   spin until my_flags[s] >= epoch for all s < W (system-scope), at most timeout_ms;
 *               on timeout status[0] = 1 + first late slot and the kernel returns (never hangs). */
int mi_ep_signal(uint64_t *const *peer_flags_host, int num_ranks, int my_rank, uint64_t epoch, void *stream);
int mi_ep_wait(const uint64_t *my_flags, int num_ranks, uint64_t epoch, int32_t *status, int timeout_ms,
               void *stream);
/* signal then wait in ONE launch (one rank per process: the peers' signals come from their own streams). */
int mi_ep_signal_wait(uint64_t *const *peer_flags_host, const uint64_t *my_flags, int num_ranks, int my_rank, uint64_t epoch,
                      uint64_t *epoch_ctr /* NULL, or: epoch = *epoch_ctr + 1, stored back when the wait is over */,
                      int32_t *status, int timeout_ms, void *stream);

/* Start-up self-test of mapped windows (the reference trusts HCCL for this; here the mapping is ours: hipIpc over xGMI).
 * `rounds` rounds on the SAME addresses, a fresh pattern each: every rank writes a 4 KiB pattern row into slot `my_rank` of every
 * rank's `peer_rows_host[d]` area (mi_ep_selftest_bytes(W) bytes each) with ordinary stores plus one {epoch, value} granule with a
 * relaxed system-scope store, raises flag `epoch`, waits for all peers (bounded), then -- with ORDINARY cached loads, in a launch of
 * its own -- verifies the W rows it received (remote-write path), its own row read back from every peer (remote-read path) and the
 * W granules, and acknowledges; round r + 1 starts writing only when every peer has acknowledged round r, so a cache line kept
 * from round r that is served stale in round r + 1 fails the test (a single round on fresh addresses cannot show that).
 * Epochs first_epoch .. first_epoch + rounds - 1 must continue where the previous call on these flag words stopped (start at 1).
 * Three launches per round on `stream`.  status[0] afterwards: 0 = pass, 1 + s = rank s never signalled, 3000 + s = corrupt row
 * from s, 4000 + d = corrupt read-back from d, 5000 + s = missing / corrupt granule from s. */
size_t mi_ep_selftest_bytes(int num_ranks);
int mi_ep_selftest(void *const *peer_rows_host, uint64_t *const *peer_flags_host, const uint64_t *my_flags,
                   uint64_t *const *peer_acks_host, const uint64_t *my_acks, int num_ranks, int my_rank, uint64_t first_epoch,
                   int rounds, uint32_t tag, int32_t *status, int timeout_ms, void *stream);
/* Second leg: the IN-LAUNCH hand-off of the two-launch low-latency forms (mi_ep_ll_dispatch_layout_send_tagged + mi_ep_ll_wait_pack,
 * mi_ep_combine_push_flagged + mi_ep_combine_reduce_flagged; reference: the per-token flag waits of moe_distribute_combine_v2.h:952-1002 and
 * moe_distribute_dispatch_v2.h:1159-1171).  One launch per round: producer workgroups write 4 KiB rows into every peer's scratch with
 * write-through stores, drain, then store the row's tag (meta word behind the payload) or raise its flag word in the peer's row-flag area;
 * consumer workgroups of the same launch poll that word (relaxed, system scope) and read the payload with system-scope loads -- the very
 * instruction sequences of those kernels.  Rounds alternate between two halves `rows_half_stride` / `flags_half_stride` bytes apart, so round
 * r + 2 revisits the addresses of round r under a fresh pattern, and every checked row is read once more with ordinary loads so that its
 * lines stay in the reader's caches.  `rounds` >= 4 covers both halves twice.  Scratch: mi_ep_selftest_inlaunch_bytes(W) bytes per half of
 * every rank's rows, mi_ep_selftest_inlaunch_flag_words(W) uint32 per half of its flag words.  Epochs continue those of mi_ep_selftest on the
 * same ack words.  skip_payload_from_round >= 0 (test hook): from that round on this rank raises tags / flags WITHOUT rewriting the
 * payload -- what a stale line looks like to its consumers.  status[0]: 0 = pass, 1 + s = rank s never arrived, 7000 + s = stale / corrupt
 * payload behind a tag from s, 7500 + s = the same behind a flag word. */
size_t mi_ep_selftest_inlaunch_bytes(int num_ranks);
size_t mi_ep_selftest_inlaunch_flag_words(int num_ranks);
int mi_ep_selftest_inlaunch(void *const *peer_rows_host, size_t rows_half_stride, uint32_t *const *peer_row_flags_host,
                            size_t flags_half_stride, uint64_t *const *peer_acks_host, const uint64_t *my_acks, int num_ranks, int my_rank,
                            uint64_t first_epoch, int rounds, uint32_t tag, int skip_payload_from_round, int32_t *status, int timeout_ms,
                            void *stream);

/* ---- A2 notify ------------------------------------------------------------------------------
 * Counts all-gather through windows.  Every rank owns `uint64_t notify[W][E+1]` granules
 * {epoch << 32 | value}; post writes row `my_rank` of every peer (E counts + its token count T),
 * wait sweeps all W*(E+1) granules until every tag == epoch and emits cnt_matrix [W, E+1] int32. */
int mi_ep_notify_post(uint64_t *const *peer_notify_host, int num_ranks, int my_rank, int num_experts,
                      const int32_t *num_tokens_per_expert, int num_tokens, uint32_t epoch, void *stream);
int mi_ep_notify_wait(const uint64_t *my_notify, int num_ranks, int num_experts, uint32_t epoch,
                      int32_t *cnt_matrix, int32_t *status, int timeout_ms, void *stream);
/* Fused forms used by the one-process-per-GPU runtime (fewer launches on the critical path):
 * notify_post_signal = mi_ep_notify_post + mi_ep_signal(peer_flags, signal_epoch);
 * notify_wait_tables = mi_ep_notify_wait + mi_ep_wait(my_flags, flag_epoch; skipped when my_flags is NULL) +
 *                      mi_ep_notify_tables, one workgroup. */
int mi_ep_notify_post_signal(uint64_t *const *peer_notify_host, uint64_t *const *peer_flags_host, int num_ranks, int my_rank,
                             int num_experts, const int32_t *num_tokens_per_expert, int num_tokens, uint32_t notify_epoch,
                             uint64_t signal_epoch, void *stream);
int mi_ep_notify_wait_tables(const uint64_t *my_notify, uint32_t notify_epoch, const uint64_t *my_flags, uint64_t flag_epoch,
                             int32_t *cnt_matrix, int num_ranks, int num_experts, int my_rank, int relative_pull,
                             int32_t *recv_count, int32_t *recv_offset, int32_t *recv_tokens_per_expert,
                             int32_t *expert_global_offset, int32_t *srcrank_in_expert_offset, int32_t *r_in_srcrank_offset,
                             int32_t *total_recv_token, int32_t *max_bs, int32_t *pull_offset, int32_t *summary_host,
                             int32_t *status, int timeout_ms, int32_t *wait_cost_stats /*[W] or NULL: += us waited per source*/,
                             void *stream);
/* Dynamic LDS of the notify_wait_tables / notify_exchange_tables workgroup for (num_ranks, num_experts): it keeps the W * (E + 1)
 * counts next to the tables' scratch.  Shapes needing more than a CU's 160 KB (0 = invalid W / E) make those two entry points
 * return MI_EP_EINVAL -- e.g. W = 64 with E = 2048; every shape with W <= 16 fits up to E = 2048, W = 64 up to E = 512. */
size_t mi_ep_notify_lds_bytes(int num_ranks, int num_experts);
/* Both halves of the exchange in ONE launch of one workgroup: post this rank's counts and its "rows staged" flag
 * (= mi_ep_notify_post_signal with signal_epoch = flag_epoch), then mi_ep_notify_wait_tables.  Used by the host runtime
 * when each rank is its own process; a harness that plays several ranks on one stream must keep the two halves apart
 * (every rank has to post before any rank can finish waiting). */
int mi_ep_notify_exchange_tables(uint64_t *const *peer_notify_host, uint64_t *const *peer_flags_host,
                                 const int32_t *num_tokens_per_expert, int num_tokens, const uint64_t *my_notify,
                                 uint32_t notify_epoch, const uint64_t *my_flags, uint64_t flag_epoch, int32_t *cnt_matrix,
                                 int num_ranks, int num_experts, int my_rank, int relative_pull, int32_t *recv_count,
                                 int32_t *recv_offset, int32_t *recv_tokens_per_expert, int32_t *expert_global_offset,
                                 int32_t *srcrank_in_expert_offset, int32_t *r_in_srcrank_offset, int32_t *total_recv_token,
                                 int32_t *max_bs, int32_t *pull_offset, int32_t *summary_host,
                                 uint64_t *epoch_ctr /* NULL, or: both epochs = *epoch_ctr + 1, stored back at the end */,
                                 size_t notify_parity_stride /* bytes between the two halves of the notify granules (peers and own) */,
                                 int32_t *status, int timeout_ms, int32_t *wait_cost_stats, void *stream);
/* Diagnose helpers (reference dispatch_wait_recv_cost_stats / combine_send_cost_stats, buffer.py:343-345,500-501): a device
 * timestamp (100 MHz ticks) and `stats[i] += microseconds since *t_start` for i < n.  Launched only when stats are requested. */
int mi_ep_timestamp(uint64_t *dst, void *stream);
int mi_ep_elapsed_add(int32_t *stats, int n, const uint64_t *t_start, void *stream);
/* Derived tables of rank `my_rank` from cnt_matrix [W, E+1] (last column = that rank's token count).
 * All int32: recv_count [L*W] (inclusive cumsum over i = le*W+src), recv_offset [L*W] (sender's
 * exclusive prefix), recv_tokens_per_expert [L], expert_global_offset [L], srcrank_in_expert_offset [L*W],
 * r_in_srcrank_offset [L*W] (zeros: round 1), total_recv_token [1], max_bs [1].
 * pull_offset [L*W]: row offset the pull kernel adds to src_base[src]; == recv_offset when
 * relative_pull == 0, recv_offset - send_prefix_src[my_rank*L] when relative_pull != 0 (per-source
 * contiguous staging, RCCL transport).  summary_host (may be NULL): pinned host int32[2 + L] =
 * {total_recv, max_bs, recv_tokens_per_expert...}: every word is written exactly once per call with a value >= 0, at system scope, in no
 * particular order -- a host that polls pre-sets every word it will read to -1 and waits for each. */
int mi_ep_notify_tables(const int32_t *cnt_matrix, int num_ranks, int num_experts, int my_rank,
                        int relative_pull, int32_t *recv_count, int32_t *recv_offset,
                        int32_t *recv_tokens_per_expert, int32_t *expert_global_offset,
                        int32_t *srcrank_in_expert_offset, int32_t *r_in_srcrank_offset,
                        int32_t *total_recv_token, int32_t *max_bs, int32_t *pull_offset,
                        int32_t *summary_host, void *stream);

/* ---- A3 normal dispatch -----------------------------------------------------------------------
 * stage: quantise (per mode) every token once and write one row per valid (t,k) into `rows`
 *   (the rank's own send window) at slot send_data_offset[e] + send_token_idx_small[t,k];
 *   row = payload | {scale, t, k, my_rank}.  x [T,H] bf16, topk_idx [T,K] int64/int32.
 * pull: for every output row r < total (= recv_count[L*W-1], read on device): find segment i,
 *   copy row pull_offset[i] + j of src_base[src] into recv_x[r] / recv_x_scales[r] / recv_src_idx[3r..].
 *   `rows_hint` sizes the grid and bounds the rows written (the capacity of the output buffers; normally >= total).
 *   recv_x_scales may be NULL for bf16. */
int mi_ep_dispatch_stage(const void *x, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                         const int32_t *send_data_offset, int num_tokens, int num_topk, int hidden,
                         int num_experts, int my_rank, int quant_mode, void *rows, void *stream);
int mi_ep_dispatch_pull(const void *const *src_base_host, const int32_t *recv_count, const int32_t *pull_offset,
                        int num_ranks, int num_local_experts, int hidden, int quant_mode, int rows_hint,
                        void *recv_x, float *recv_x_scales, int32_t *recv_src_idx, void *stream);
/* Compact staging for the pull transport (what the host runtime uses in normal mode).  A token selected by K experts is
 * quantised and written ONCE -- row t of `region` = payload | {scale, t, 0, my_rank} -- instead of once per (t, k); the
 * expert-sorted order travels as an index of K * 8 bytes per token: entry send_data_offset[e] + send_token_idx_small[t,k]
 * = {t, k} (u32 pairs) at byte mi_ep_dispatch_index_offset(...) of the region.  Staging writes T*(H+16) + T*K*8 bytes
 * instead of T*K*(H+16) (rows MI_EP_ROW_STRIDE apart); the received rows, scales and (src, t, k) triples are identical to stage + pull
 * (reference contract: cam_moe_dispatch_normal.h:717-760).  `region_bytes` (the same on every rank) fixes the index offset
 * and the token capacity, index_offset / row_bytes; stage_compact returns MI_EP_EINVAL when T exceeds it.
 * pull_indexed: row r of segment i = (le, src), position j, is token row index[pull_offset[i] + j].t of src_base[src]. */
size_t mi_ep_dispatch_index_offset(int hidden, int quant_mode, int num_topk, size_t region_bytes);
int mi_ep_dispatch_stage_compact(const void *x, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                                 const int32_t *send_data_offset, int num_tokens, int num_topk, int hidden, int num_experts,
                                 int my_rank, int quant_mode, void *region, size_t region_bytes,
                                 const uint64_t *epoch_ctr, size_t parity_stride /* stage side: half (*epoch_ctr + 1) & 1 */, void *stream);
int mi_ep_dispatch_pull_indexed(const void *const *src_base_host, const int32_t *recv_count, const int32_t *pull_offset,
                                int num_ranks, int num_local_experts, int hidden, int num_topk, int quant_mode, int rows_hint,
                                size_t region_bytes, void *recv_x, float *recv_x_scales, int32_t *recv_src_idx,
                                const uint64_t *epoch_ctr, size_t parity_stride /* consume side: half *epoch_ctr & 1 */,
                                int skip_src /* -1, or a source rank whose rows mi_ep_dispatch_pull_local writes */, void *stream);
/* pull_indexed without the copy (fused_deep_moe, prefill sizes): for every receive row r < rows_cap the byte offset of its STAGED row from
 * *a_base_out (the lowest source base; the ping-pong half is inside the offset), its scale and its (src, t, k) triple -- what
 * mi_ep_moe_gemm1_swiglu_rows multiplies in place.  All sources must be local memory (push transport, or num_ranks == 1); MI_EP_ESIZE when
 * a staged row could lie 4 GiB or more above the lowest base (the caller then gathers with pull_local / pull_indexed as before).
 * Reference: the MIX kernel multiplies rows where the dispatch left them, csrc/deepep/ops/op_kernel/fused_deep_moe.h:336-427. */
int mi_ep_dispatch_resolve_rows(const void *const *src_base_host, const int32_t *recv_count, const int32_t *pull_offset, int num_ranks,
                                int num_local_experts, int hidden, int num_topk, int quant_mode, int rows_cap, size_t region_bytes,
                                const void **a_base_out, uint32_t *row_offsets, float *recv_x_scales, int32_t *recv_src_idx,
                                const uint64_t *epoch_ctr, size_t parity_stride /* consume side: half *epoch_ctr & 1 */, void *stream);
/* The rows whose token lives on this rank, token by token: the staged row of token t (`my_rows` = the own region for the pull
 * transport, the own source slab for push; half 0) is read once and stored to each selection (t, k) served by this rank's experts,
 * at output row recv_count[le * W + me] - num_tokens_per_expert[me * L + le] + send_token_idx_small[t, k] -- the rows, scales and
 * triples mi_ep_dispatch_pull_indexed writes for source `my_rank`, which is then called with skip_src = my_rank (not at all when
 * num_ranks == 1).  K-fold fewer reads of the staged rows for the local share of the traffic.
 * local_row_out (NULL or int32 [T * K]): entry t * K + k receives the output row of every selection written here -- exactly the
 * `local_row` table mi_ep_combine_push builds for the rows of own-rank tokens, available a whole expert computation earlier; at
 * num_ranks == 1 it covers every row, and the combine of that exchange is mi_ep_combine_reduce alone (no push, no signal / wait). */
int mi_ep_dispatch_pull_local(const void *my_rows, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                              const int32_t *recv_count, const int32_t *num_tokens_per_expert, int num_tokens, int num_topk,
                              int hidden, int num_experts, int num_ranks, int my_rank, int quant_mode, int rows_hint, void *recv_x,
                              float *recv_x_scales, int32_t *recv_src_idx, int32_t *local_row_out, const uint64_t *epoch_ctr,
                              size_t parity_stride, void *stream);

/* Push transport of normal dispatch (selectable next to the pull above; the same received bytes).  The sender writes the
 * quantised row of token t ONCE into the window of every rank that owns at least one of the token's experts (the reference
 * also writes into the receiver's window, cam_moe_dispatch_normal.h:440-473, but one row per (t, k)): the dispatch region of a
 * rank is cut into W source slabs of mi_ep_dispatch_push_slab_bytes() and rank s writes only slab s -- token row t at row t
 * of the slab (sparse: only the tokens routed there), and for every pair (t, k) routed there the index entry {t, k} at
 * position send_data_offset[e] - send_data_offset[first expert of that rank] + send_token_idx_small[t,k], i.e. in the
 * sender's expert-sorted order.  Nothing depends on the other ranks' counts, so the push runs before the notify exchange;
 * cross-GPU traffic is distinct (token, destination rank) pairs * (H + 16) bytes + 8 bytes per pair.
 * Receiver: after the notify exchange computed with relative_pull = 1, mi_ep_dispatch_pull_indexed with
 * src_base_host[s] = own region + s * slab and region_bytes = the slab size gathers recv_x / scales / triples locally.
 * peer_region_host[W]: every rank's dispatch region for this call.  MI_EP_EINVAL when T exceeds a slab's token capacity. */
size_t mi_ep_dispatch_push_slab_bytes(size_t region_bytes, int num_ranks);
int mi_ep_dispatch_stage_push(const void *x, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                              const int32_t *send_data_offset, int num_tokens, int num_topk, int hidden, int num_experts,
                              int num_ranks, int my_rank, int quant_mode, void *const *peer_region_host, size_t region_bytes,
                              const uint64_t *epoch_ctr, size_t parity_stride, void *stream);

/* ---- A4/A6 combine ---------------------------------------------------------------------------
 * push: row r < total (= *total_rows_dev if non-NULL else rows_hint) of x [R,H] bf16 with triple
 *   (src, t, k) = src_idx[3r..3r+2] is copied to dst_base[src] + (t*K + k) * mi_ep_combine_row_bytes(H).
 * reduce: out[t] = bf16_rne( sum_{k asc, 0 <= idx[t,k] < E} float(slot[t*K+k]) * w[t,k] ) with separate fp32
 *   multiply and add (cam_moe_combine_normal.h:372-396).  topk_weights NULL -> ones.
 *   send_data_offset / send_token_idx_small: both NULL for the window layout above. */
/* push: rows >= min(*total_rows_dev, rows_hint) are not touched; a triple outside [0,W) x slots of `slot_region_bytes`
 * (0 = unchecked) x [0,K) is dropped instead of becoming a wild cross-GPU store. */
/* Rows that do not travel (local_row != NULL on the push, x_local != NULL on the reduce): a row whose token lives on this rank
 * (src == my_rank) is not copied into the window; the push stores its row number r at local_row[t*K + k] and the reduce reads
 * selection (t, k) -- expert idx[t,k] / (num_experts / num_ranks) == my_rank -- from x_local [local_rows, H] row local_row[t*K + k]
 * (clamped into x_local).  Same values, same k-ascending order: bit-identical to the all-through-the-window path; saves the read
 * + write of the push and reads the same bytes in the reduce (all of the combine traffic at EP = 1, 1/W of it at EP = W).
 * Not available with the all-to-all slot layout (send_data_offset != NULL). */
int mi_ep_combine_push(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint,
                       int hidden, int num_topk, void *const *dst_base_host, int num_ranks, size_t slot_region_bytes,
                       const uint64_t *epoch_ctr, size_t parity_stride, int my_rank, int32_t *local_row, void *stream);
/* push + "my rows are pushed" + the wait for every expert rank in ONE launch (mi_ep_combine_push followed by mi_ep_signal_wait with
 * epoch_ctr): every workgroup writes its rows through the caches, drains and counts itself in at *arrive_word (a device word of the
 * rank's own control area: zero before the first call; the kernel re-arms it); the last workgroup to arrive raises this rank's flag at
 * every owner, waits (bounded by timeout_ms, reported through *status) for every expert rank's and completes the family's call counter.
 * Same bytes in the same slots.  One call in flight per arrive_word. */
int mi_ep_combine_push_signal_wait(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int hidden, int num_topk,
                                   void *const *dst_base_host, int num_ranks, size_t slot_region_bytes, uint64_t *epoch_ctr,
                                   size_t parity_stride, int my_rank, int32_t *local_row, uint64_t *const *peer_flags_host,
                                   const uint64_t *my_flags, uint32_t *arrive_word, int32_t *status, int timeout_ms, void *stream);
/* The TWO-launch combine (the reference waits per token for the rows it sums, moe_distribute_combine_v2.h:952-1002): mi_ep_combine_push_flagged =
 * mi_ep_combine_push whose waves write their rows through the caches and, once a row's stores have drained, raise the row's flag word at
 * its owner -- uint32 [2 halves, row_flags_parity_stride bytes apart][slot rows] in every rank's control area, word t * K + k of the half
 * (*epoch_ctr + 1) & 1, value = the low 32 bits of that epoch (tags, never cleared); mi_ep_combine_reduce_flagged = mi_ep_combine_reduce
 * whose waves wait (bounded by timeout_ms, code 3000 + k through *status) for the words of their token's valid, non-local selections before
 * they read the rows.  The push leaves the call's epoch at *cur_epoch_word (a word of the rank's own control area, the same for both calls; it
 * is launched even without rows), the reduce takes epoch and ping-pong half from there and completes *epoch_ctr.
 * No "rows pushed" flag exchange, no single-workgroup launch between the two; same bytes in the same slots, same sum.
 * max_blocks (here and in mi_ep_ll_wait_pack): 0, or a cap on the workgroups of the WAITING launch -- for ranks that share one GPU (test setups):
 * two waiting workgroups of one process on every CU kept another process's 1024-thread send workgroups from starting (64 is safe). */
int mi_ep_combine_push_flagged(const void *x, const int32_t *src_idx, const int32_t *total_rows_dev, int rows_hint, int hidden, int num_topk,
                               void *const *dst_base_host, int num_ranks, size_t slot_region_bytes, const uint64_t *epoch_ctr,
                               size_t parity_stride, int my_rank, int32_t *local_row, uint32_t *const *peer_row_flags_host,
                               size_t row_flags_parity_stride, uint64_t *cur_epoch_word, void *stream);
int mi_ep_combine_reduce_flagged(const void *slots, const void *topk_idx, int idx_is_i32, const float *topk_weights, int num_tokens, int num_topk,
                                 int hidden, int num_experts, void *out, uint64_t *epoch_ctr, size_t parity_stride, const void *x_local,
                                 const int32_t *local_row, int local_rows, int my_rank, int num_ranks, const uint32_t *my_row_flags,
                                 size_t row_flags_parity_stride, const uint64_t *cur_epoch_word, int32_t *status, int timeout_ms, int max_blocks,
                                 void *stream);
int mi_ep_combine_reduce(const void *slots, const void *topk_idx, int idx_is_i32, const float *topk_weights,
                         const int32_t *send_data_offset, const int32_t *send_token_idx_small, int num_tokens,
                         int num_topk, int hidden, int num_experts, void *out, const uint64_t *epoch_ctr, size_t parity_stride,
                         const void *x_local, const int32_t *local_row, int local_rows, int my_rank, int num_ranks, void *stream);
/* All-to-all (RCCL) transport helper: reorder x [R,H] bf16 from dispatch order (local expert, src, j) into
 * per-source blocks (src, local expert, j) -- each block is what that source staged for this rank, in its
 * send-slot order, so it can be returned as one contiguous message.  send_head [L*W] = recv_count of the
 * dispatch.  rows_per_src [W] (may be NULL) receives the block sizes.  With this transport the reducer reads
 * slot send_data_offset[e] + send_token_idx_small[t,k] (pass both to mi_ep_combine_reduce) instead of t*K+k. */
int mi_ep_combine_pack(const void *x, const int32_t *send_head, int num_ranks, int num_local_experts, int hidden,
                       int rows_hint, void *packed, int32_t *rows_per_src, void *stream);

/* ---- shared-expert ranks (MOE_SHARED_EXPERT_RANK_NUM = S > 0; reference deep_ep.cpp:62,866-874,1219-1220) --------------------------
 * The first S ranks hold ONE expert each (the shared expert), the other W - S ranks L = num_moe_experts / (W - S) routed experts each;
 * every token with at least one active selection also goes to shared rank (my_rank mod S) at its position among those tokens, triple
 * k = K (moe_distribute_dispatch_v2.h:555-604,748-779,918-960), and the combine adds that row unweighted behind the K weighted ones
 * (moe_distribute_combine_v2.h:1219-1235).  This call renames the experts so that the ordinary kernels do exactly that with
 * num_experts = W * L and num_topk = K + 1: idx_out [T, K+1] int32 -- routed expert e -> S*L + e, the shared selection (my_rank mod S) * L
 * (or -1 for a token without an active selection), invalid ids -> -1; weights_out [T, K+1] (NULL: not needed) = topk_weights (NULL: ones)
 * with 1.0f in the last column (x * 1.0f is exact: the same bits as the reference's plain add).  Shared ranks use local slot 0 only:
 * their packed rows, counts [0, W) and row total are those of the reference's single local expert. */
int mi_ep_shared_expert_map(const void *topk_idx, int idx_is_i32, const float *topk_weights, int num_tokens, int num_topk,
                            int num_moe_experts, int num_ranks, int shared_ranks, int my_rank, int32_t *idx_out, float *weights_out,
                            void *stream);

/* ---- A5 low-latency dispatch -------------------------------------------------------------------
 * Window of a rank: rows [L][W][max_tokens] of mi_ep_dispatch_row_bytes(), counts granules
 * uint64 [L*W] {epoch<<32|count}.
 * send: quantise + write the row of every valid (t,k) straight into the destination rank's region
 *   (le, my_rank) at position send_token_idx_small[t,k].
 * post_counts: after `send` (stream order) publish count[le][my_rank] granules to every destination.
 * recv: sweep my L*W granules (bounded spin) -> layout_range [L*W] inclusive cumsum, packed_recv_count [L]
 *   int64 (count, or cumulative when count_type == 0), then compact window rows into packed_recv_x /
 *   packed_recv_x_scales / src_info triples in idx-i order. */
int mi_ep_ll_dispatch_send(const void *x, const void *topk_idx, int idx_is_i32, const int32_t *send_token_idx_small,
                           int num_tokens, int num_topk, int hidden, int num_experts, int num_ranks, int my_rank,
                           int max_tokens, int quant_mode, void *const *peer_rows_host, const uint64_t *epoch_ctr,
                           size_t parity_stride, void *stream);
/* mi_ep_dispatch_layout + mi_ep_ll_dispatch_send in ONE launch (the form the host runtime uses for low-latency dispatch, <= 1024
 * tokens, num_experts <= 1024): the first workgroup computes the layout tables of the batch (all five outputs are written, as by
 * mi_ep_dispatch_layout: the count exchange needs num_tokens_per_expert); the other workgroups are the send waves, one per
 * (token, selection), which find their slab position themselves -- the number of earlier (t, k) pairs with the same expert, counted in
 * an LDS copy of the routing table -- so no workgroup waits for another one.  Same rows, tables and bytes as the two separate calls. */
int mi_ep_ll_dispatch_layout_send(const void *x, const void *topk_idx, int idx_is_i32, int num_tokens, int num_topk, int hidden,
                                  int num_experts, int num_ranks, int my_rank, int max_tokens, int quant_mode,
                                  void *const *peer_rows_host, const uint64_t *epoch_ctr, size_t parity_stride,
                                  int32_t *num_tokens_per_rank, int32_t *num_tokens_per_expert, int32_t *is_token_in_rank,
                                  int32_t *send_token_idx_small, int32_t *send_data_offset, void *stream);
int mi_ep_ll_post_counts(uint64_t *const *peer_counts_host, const int32_t *num_tokens_per_expert, int num_experts,
                         int num_ranks, int my_rank, uint32_t epoch, void *stream);
int mi_ep_ll_dispatch_recv(const void *my_rows, const uint64_t *my_counts, uint32_t epoch, int num_ranks,
                           int num_local_experts, int max_tokens, int hidden, int quant_mode, int count_type,
                           void *packed_recv_x, float *packed_recv_x_scales, int64_t *packed_recv_count,
                           int32_t *src_info, int32_t *layout_range, int rows_capacity /* rows packed_recv_x holds; 0 = L*W*max_tokens */,
                           int32_t *status, int timeout_ms, void *stream);
/* Layout + send + the count exchange in ONE launch, then the packing launch: two launches per low-latency dispatch instead of three.
 * mi_ep_ll_dispatch_layout_send_counts = mi_ep_ll_dispatch_layout_send whose workgroups count themselves in at *arrive_word (a device
 * word of the rank's own control area: zero before the first call; the kernel re-arms it) once their rows -- written through the caches
 * -- have drained; the last one to arrive posts this rank's per-expert counts, collects everybody's (bounded by timeout_ms, reported
 * through *status), leaves the cumulative counts in layout_range [L*W] and packed_recv_count [L], and completes the family's call
 * counter (epoch_ctr: required).  mi_ep_ll_pack = the packing half of mi_ep_ll_post_recv.  Same rows, tables and bytes. */
int mi_ep_ll_dispatch_layout_send_counts(const void *x, const void *topk_idx, int idx_is_i32, int num_tokens, int num_topk, int hidden,
                                         int num_experts, int num_ranks, int my_rank, int max_tokens, int quant_mode, void *const *peer_rows_host,
                                         uint64_t *epoch_ctr, size_t parity_stride, int32_t *num_tokens_per_rank,
                                         int32_t *num_tokens_per_expert, int32_t *is_token_in_rank, int32_t *send_token_idx_small,
                                         int32_t *send_data_offset, uint64_t *const *peer_counts_host, const uint64_t *my_counts,
                                         size_t counts_parity_stride, int count_type, int32_t *layout_range, int64_t *packed_recv_count,
                                         uint32_t *arrive_word, int32_t *status, int timeout_ms, void *stream);
/* The TWO-launch low-latency dispatch with nothing between the launches.  mi_ep_ll_dispatch_layout_send_tagged = mi_ep_ll_dispatch_layout_send
 * whose layout workgroup posts this rank's per-expert counts to every peer's count granules as soon as it has them (half (*epoch_ctr + 1) & 1,
 * granule = epoch << 32 | count) and leaves the call's epoch at *cur_epoch_word (own control area), and whose send waves write every row's
 * payload through the caches, drain, and only then its meta -- word 3 = src_rank | tag << 8, tag = epoch % 0xFFFFFF + 1: nobody waits in this
 * launch.  mi_ep_ll_wait_pack = the count exchange's collecting half + mi_ep_ll_pack in one launch of <= 512 workgroups: each collects the L * W
 * granules of the epoch at *cur_epoch_word (bounded by timeout_ms; code 2000 + i through *status), scans them, and packs its rows, waiting
 * per row for the tag (code 2500 + ..); workgroup 0 writes layout_range / packed_recv_count and completes *epoch_ctr.  Same rows, tables and
 * bytes as the three-launch form (src_info word 0 = the meta word's low 8 bits).  Reference: per-token arrival state instead of a global
 * barrier, moe_distribute_dispatch_v2.h:1477-1490. */
int mi_ep_ll_dispatch_layout_send_tagged(const void *x, const void *topk_idx, int idx_is_i32, int num_tokens, int num_topk, int hidden,
                                         int num_experts, int num_ranks, int my_rank, int max_tokens, int quant_mode, void *const *peer_rows_host,
                                         const uint64_t *epoch_ctr, size_t parity_stride, int32_t *num_tokens_per_rank,
                                         int32_t *num_tokens_per_expert, int32_t *is_token_in_rank, int32_t *send_token_idx_small,
                                         int32_t *send_data_offset, uint64_t *const *peer_counts_host, size_t counts_parity_stride,
                                         uint64_t *cur_epoch_word, void *stream);
int mi_ep_ll_wait_pack(const void *my_rows, const uint64_t *my_counts, size_t counts_parity_stride, int num_ranks, int num_local_experts,
                       int max_tokens, int hidden, int quant_mode, int count_type, void *packed_recv_x, float *packed_recv_x_scales,
                       int64_t *packed_recv_count, int32_t *src_info, int32_t *layout_range, int rows_capacity, const uint64_t *cur_epoch_word,
                       uint64_t *epoch_ctr, size_t rows_parity_stride, int32_t *status, int timeout_ms, int max_blocks, void *stream);
int mi_ep_ll_pack(const void *my_rows, const int32_t *layout_range, int num_ranks, int num_local_experts, int max_tokens, int hidden,
                  int quant_mode, void *packed_recv_x, float *packed_recv_x_scales, int32_t *src_info, int rows_capacity,
                  const uint64_t *epoch_ctr, size_t rows_parity_stride, void *stream);
/* ll_post_counts fused into the receive (one rank per process): the counts workgroup posts this rank's counts, collects
 * everybody's, scans them; the packing kernel follows.  Two launches instead of three. */
int mi_ep_ll_post_recv(uint64_t *const *peer_counts_host, const int32_t *num_tokens_per_expert, int my_rank, const void *my_rows,
                       const uint64_t *my_counts, uint32_t epoch, int num_ranks, int num_local_experts, int max_tokens, int hidden,
                       int quant_mode, int count_type, void *packed_recv_x, float *packed_recv_x_scales,
                       int64_t *packed_recv_count, int32_t *src_info, int32_t *layout_range, int rows_capacity,
                       uint64_t *epoch_ctr, size_t rows_parity_stride, size_t counts_parity_stride, int32_t *status, int timeout_ms,
                       void *stream);

/* ---- A8 fused_deep_moe building blocks (reference: aclnnFusedDeepMoe, csrc/deepep/deep_ep.cpp:1223;
 * kernel csrc/deepep/ops/op_kernel/fused_deep_moe.h:336-427) -------------------------------------------------------
 * Rows are the packed output of the low-latency dispatch; expert e owns rows [cum[e*s-1], cum[(e+1)*s-1]) of the
 * inclusive cumulative table `row_cumsum` (s = cum_stride; pass layout_range with s = W).  rows_cap bounds the rows.
 * gemm1_swiglu: a int8 [rows, hidden] (per-token scales a_scale), w int8 [L, two_i, hidden] (K contiguous; columns of every
 *   128-wide tile are 64 gate then 64 up, i.e. reshape_fusion_gmm_weight of tests/python/deepep/test_fused_deep_moe.py:75-86),
 *   w_scale f32 [L, two_i] permuted alike.  out f32 [rows, two_i/2]: up * silu(gate), gate/up = (i32 * w_scale[col]) * a_scale[row].
 * rowquant: q = int8 rint((v * 127) * (1 / rowmax)), scale = rowmax / 127 for rows < *total_rows_dev.
 * gemm2: a int8 [rows, inter] with scales, w int8 [L, hidden, inter], w_scale f32 [L, hidden]; out bf16 [rows, hidden].
 * hidden, inter multiples of 128; two_i multiple of 128.  rows_per_expert_hint: expected rows per local expert (0 = unknown);
 * it only selects the tile shape (<= 96: 64-row tiles for decode-size groups, else 256-row tiles). */
int mi_ep_moe_gemm1_swiglu(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale,
                           const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap, int hidden,
                           int two_i, float *out, int rows_per_expert_hint, void *stream);
/* gemm1_swiglu over rows that were never gathered: row r of the packed order is the `hidden` int8 bytes at a_base + a_row_offsets[r]
 * (16-byte aligned byte offsets below 4 GiB; rows >= the cumulative total are not read) -- the staged token rows of a dispatch, one per
 * TOKEN and shared by its K selections, addressed through mi_ep_dispatch_resolve_rows' table.  Same products, same outputs. */
int mi_ep_moe_gemm1_swiglu_rows(const void *a_base, const uint32_t *a_row_offsets, const float *a_scale, const int8_t *w,
                                const float *w_scale, const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap,
                                int hidden, int two_i, float *out, int rows_per_expert_hint, void *stream);
/* gemm1_swiglu + rowquant in ONE launch (the reference requantises in GEMM1's epilogue as well: block_epilogue_per_token_dequant_swiglu.h:250-269,
 * grouped_matmul_slice_m_per_token_dequant_swiglu_quant_multistage_workspace.h:199-265): the two_i / 256 column-tile workgroups of a row block
 * exchange their rows' maxima (one atomicMax per row and workgroup + an arrival count, bounded wait) and quantise their SwiGLU values from
 * registers -- q and q_scale carry the bits mi_ep_moe_gemm1_swiglu + mi_ep_moe_rowquant produce, no fp32 [rows, inter] intermediate exists.
 * a_row_offsets NULL: a_base is the dense int8 [rows, hidden]; else as mi_ep_moe_gemm1_swiglu_rows.  two_i must be a multiple of 256.
 * zeroed_words: mi_ep_moe_requant_words(rows_cap, L) uint32 that are ZERO when the launch starts (one memset per call).  xcds: the value
 * mi_ep_moe_probe_xcds() returned on this device (how workgroups are dealt to XCDs; 1 assumes nothing about placement).  status: a
 * device-visible word, MI_EP_STATUS_GEMM_ROWMAX is stored there if a row block's column tiles did not all arrive within timeout_ms. */
#define MI_EP_STATUS_GEMM_ROWMAX 9000
size_t mi_ep_moe_requant_words(int rows_cap, int num_local_experts);
int mi_ep_moe_probe_xcds(void *stream);
int mi_ep_moe_gemm1_swiglu_quant(const void *a_base, const uint32_t *a_row_offsets, const float *a_scale, const int8_t *w,
                                 const float *w_scale, const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap,
                                 int hidden, int two_i, int8_t *q, float *q_scale, uint32_t *zeroed_words, int xcds, int32_t *status,
                                 int timeout_ms, int rows_per_expert_hint, void *stream);
int mi_ep_moe_rowquant(const float *v, const int32_t *total_rows_dev, int rows_cap, int inter, int8_t *q, float *scale,
                       void *stream);
int mi_ep_moe_gemm2(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale, const int32_t *row_cumsum,
                    int cum_stride, int num_local_experts, int rows_cap, int inter, int hidden, void *out_bf16,
                    int rows_per_expert_hint, void *stream);
/* GEMM2 with the combine push fused into its epilogue: the bf16 row r is not written to a dense [rows, hidden] tensor but
 * straight into slot t*topk+k of rank src's combine window, (src, t, k) = src_idx[3 r ..] -- the same destination and the
 * same bytes as mi_ep_moe_gemm2 followed by mi_ep_combine_push (the reference's fused op also sends from its GEMM2
 * epilogue: fused_deep_moe.h:336-427).  dst_base_host[W] = every rank's combine region for this call (host array). */
/* Shader clock the chip held under the LAST launch of each grouped-GEMM form on the current device (synchronising read of three device
 * words; a measurement aid): ghz3 / us3 [0] = GEMM1 + SwiGLU, [1] = GEMM2, [2] = GEMM2 + push: shader-clock ticks over 100 MHz reference
 * ticks of the first workgroup's run, and that run's duration.  0 where the form has not run. */
int mi_ep_moe_gemm_clock(double *ghz3, double *us3);
int mi_ep_moe_gemm2_push(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale, const int32_t *row_cumsum,
                         int cum_stride, int num_local_experts, int rows_cap, int inter, int hidden, const int32_t *src_idx,
                         int topk, void *const *dst_base_host, int num_ranks, size_t slot_region_bytes, const uint64_t *epoch_ctr,
                         size_t parity_stride, int rows_per_expert_hint, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_EP_H_ */
