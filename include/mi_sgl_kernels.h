/*
 * mi_sgl_kernels.h -- C-ABI of the MI355X (gfx950) fused inference primitives that sit either side of the
 * DeepEP dispatch/combine path: paged MLA / GQA decode attention, SwiGLU + INT8 quantisation, fused Add+RMSNorm
 * (+bias, +static INT8 quantisation), split-QKV RMSNorm + RoPE.
 *
 * These replace the reference's Triton-Ascend kernels (python/sgl_kernel_npu/sgl_kernel_npu/...):
 *   mi_mla_decode            <- attention/decode_attention.py:5-230   (_paged_mla_fwd_kernel / decode_mla)
 *   mi_gqa_decode            <- attention/decode_attention.py:233-450,646-760 (decode_gqa, decode_gqa_high_performance)
 *   mi_gqa_decode_sinks      <- attention/sinks_attention.py:7-286 (attention_sinks_triton, attention_sinks_prefill_triton)
 *   mi_swiglu_quant          <- activation/swiglu_quant.py:8-127      (_swiglu_quant_kernel / swiglu_quant)
 *   mi_add_rmsnorm_bias      <- norm/add_rmsnorm_bias.py:8-147        (add_rmsnorm_bias_kernel / add_rmsnorm_bias)
 *                               norm/add_rmsnorm_bias.py:150-232      (add_gemma_rms_norm)
 *   mi_split_qkv_rmsnorm_rope<- norm/split_qkv_rmsnorm_rope.py:8-438  (split_qkv_rmsnorm_rope)
 *   mi_split_qkvgate_gemma_rmsnorm_rope <- norm/split_qkv_rmsnorm_rope.py:441-745  (split_qkvgate_gemma_rmsnorm_rope)
 *   mi_rope_qk_mqa           <- norm/fused_rope_qk_mqa.py:6-160       (fused_rope_qk_mqa)
 * and are what `torch.ops.npu.*` (csrc/pytorch_extensions.cpp) and the `sgl_kernel_npu` Python functions bind.
 *
 * Conventions: plain DEVICE pointers and sizes; every call only enqueues work on `stream` (hipStream_t as void*),
 * never allocates or synchronises; returns 0 on success, negative on bad arguments / launch failure.
 */
#ifndef MI_SGL_KERNELS_H_
#define MI_SGL_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_SGL_OK 0
#define MI_SGL_EINVAL (-1)
#define MI_SGL_ELAUNCH (-2)
#define MI_SGL_ENOTAPPLICABLE (-3)      /* a fast path that does not serve this shape: take the general one */

#define MI_DTYPE_BF16 0
#define MI_DTYPE_F16 1
#define MI_DTYPE_F32 2      /* the row statistics / scalings below only */

const char *mi_sgl_kernels_version(void);

/* ---- paged MLA decode attention ----------------------------------------------------------------------------
 * out[b,h,:] = softmax_n( (q_nope[b,h]·k_nope[n] + q_rope[b,h]·k_rope[n]) * sm_scale ) · k_nope[n],  n < kv_seq_lens[b]
 * (V aliases K_nope; fp32 scores / softmax / accumulation, P rounded to the KV dtype before P·V, as in the
 *  reference kernel decode_attention.py:110-163).
 *   q          [batch, q_heads, 576]            strides (q_stride_b, q_stride_h) in elements, last dim contiguous
 *   k_nope     [num_blocks, page_size, kv_heads, 512]   strides (kn_stride_blk, kn_stride_row, kn_stride_h)
 *   k_rope     [num_blocks, page_size, kv_heads, 64]    strides (kr_stride_blk, kr_stride_row, kr_stride_h)
 *   out        [batch, q_heads, 512]            strides (o_stride_b, o_stride_h)
 *   kv_seq_lens int32 [batch]; block_table int32 [batch, bt_stride] (logical page -> physical block)
 * q_heads % kv_heads == 0; any page_size >= 1.  num_splits >= 1 partitions the KV range of every sequence
 * (flash-decoding).  `workspace` of mi_mla_decode_workspace() bytes (device memory, contents irrelevant) is needed when
 * num_splits > 1 or q_heads / kv_heads > 64.  Pass num_splits = 0 to let the library choose (mi_mla_decode_num_splits).
 * MI_MLA_SPLITS_PLANNED (what mi_mla_decode_num_splits returns for kv groups of up to 128 heads): no uniform split count at all -- one small
 * launch in front of the kernel reads kv_seq_lens ON THE DEVICE (no host sync) and cuts every sequence into pieces of about
 * (all tiles + fixed costs) / CUs, longest first; a batch of ragged lengths then runs at the pace of the average sequence, not of
 * its longest.  Pass the same value to mi_mla_decode_workspace. */
#define MI_MLA_SPLITS_PLANNED (-1)
size_t mi_mla_decode_workspace(int batch, int q_heads, int num_splits);
/* Plan once, run many: the work list of the planned form depends only on kv_seq_lens (contents), batch and kv_heads -- the attention layers
 * of one decode step share it.  mi_mla_decode_build_plan: one small launch into caller memory of mi_mla_decode_plan_bytes();
 * mi_mla_decode_with_plan = mi_mla_decode(num_splits = MI_MLA_SPLITS_PLANNED) without that launch (workspace as for that value).  HARD
 * requirements on the plan: built for the SAME batch * kv_heads (the item offsets inside the list depend on that count and are not
 * checked by the kernels) on the same device (same worker count).  Soft: the kv_seq_lens contents -- every consumer clamps its piece to
 * the tiles its sequence has NOW and the last piece of a sequence runs to their end, so a plan built from other lengths costs balance,
 * never correctness.  MI_SGL_ENOTAPPLICABLE where the
 * planned form does not serve the shape (kv groups of more than 128 heads; for groups of 65..128 heads, page sizes that are not powers of
 * two): call mi_mla_decode.  One list serves every head count: it counts 32-key tiles and its pieces start on even tiles. */
size_t mi_mla_decode_plan_bytes(int batch, int kv_heads);
int mi_mla_decode_build_plan(const int32_t *kv_seq_lens, int batch, int kv_heads, void *plan, size_t plan_bytes, void *stream);
int mi_mla_decode_with_plan(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                            const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                            int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk, int64_t kn_stride_row,
                            int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row, int64_t kr_stride_h,
                            int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, const void *plan, void *workspace,
                            size_t workspace_bytes, void *stream);
/* Introspection of the planned form (tests, tuning): byte offset of the work list inside a workspace sized with MI_MLA_SPLITS_PLANNED, and
 * the number of concurrently running workgroups it was balanced for (the CU count).  Work-list words (int32):
 *   [0] items, [1] piece size (tiles);  [16 + 2 s] first item of (sequence, kv head) pair s, [17 + 2 s] its piece count n_s;  then 4 words
 *   per item: pair (-1 = behind the list), first tile, end tile, k | n_s << 8 (tiles of 32 keys, pieces start on even tiles).  The pieces of a
 *   pair are consecutive items (item of piece k: first + k), pairs in order of descending length: any run of items spreads evenly over the
 *   XCDs (workgroup i runs on XCD i mod 8). */
size_t mi_mla_decode_plan_offset(int batch, int q_heads);
int mi_mla_decode_plan_workers(void);
/* NOTE (API change since 0.1-r4): returns MI_MLA_SPLITS_PLANNED (-1) for kv groups of up to 128 heads -- a value to pass back to
 * mi_mla_decode / mi_mla_decode_workspace, NOT a positive count to loop over; any other negative num_splits is MI_SGL_EINVAL.  Callers
 * that need a positive uniform count call mi_mla_decode_uniform_splits. */
int mi_mla_decode_num_splits(int batch, int q_heads, int kv_heads, int max_seq_len);
int mi_mla_decode_uniform_splits(int batch, int q_heads, int kv_heads, int max_seq_len);
/* kv groups of more than 64 heads have two kernel forms: 8 waves per workgroup (two per SIMD, the default) and 4 (one per SIMD).
 * waves = 4 / 8 forces one for the calls that follow, 0 returns to the default (or MI_MLA_WIDE8).  Process-wide, not thread-safe:
 * a test / tuning knob, results do not depend on it beyond fp32 summation order. */
int mi_mla_decode_select_wide(int waves);
/* Sequences the work list (or num_splits = 2) cuts in exactly TWO pieces finish between their two workgroups: each publishes its partial,
 * waits (bounded) for the other's and writes one half of the output dimensions itself -- the merge launch finds nothing to do for them
 * (BASELINE C4: batch 128 on 256 CUs).  Same sums in the same order as the merge kernel: bit-identical outputs.  mode 0 / 1 = off / on
 * for the calls that follow, -1 = default (on; MI_MLA_PAIR=0 in the environment turns it off), 2 = on with the second piece withholding
 * its word, so the first runs into its bounded wait and the merge kernel does the work (a test of that path).  Process-wide knob. */
int mi_mla_decode_set_pair(int mode);
int mi_mla_decode(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                  const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                  int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk, int64_t kn_stride_row,
                  int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row, int64_t kr_stride_h,
                  int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, int num_splits, void *workspace,
                  size_t workspace_bytes, void *stream);

/* ---- paged GQA decode attention with a separate V cache (decode_attention.py:233-450) -----------------------------
 * out[b,h,:] = softmax_n( q[b,h]·k[n] * sm_scale ) · v[n],  n < kv_seq_lens[b]; kv head of query head h = h / (q_heads / kv_heads).
 *   q [batch, q_heads, k_dim]; k [num_blocks, page_size, kv_heads, k_dim]; v [num_blocks, page_size, kv_heads, v_dim]
 *   (v may alias k); out [batch, q_heads, v_dim]; strides in elements, last dims contiguous.
 * k_dim, v_dim multiples of 8 with (k_dim, v_dim) <= one of (64,64) (128,128) (192,128) (256,256) (288,256) (576,512).
 * num_splits as for mi_mla_decode. */
size_t mi_gqa_decode_workspace(int batch, int q_heads, int v_dim, int num_splits);
int mi_gqa_decode_num_splits(int batch, int q_heads, int kv_heads, int max_seq_len);
/* Large kv groups (65..128 query heads per kv head, head dims <= (288, 256): the kernel of gqa_decode_wide.hip): sequences cut in exactly TWO
 * pieces (num_splits = 2, or by the work list) finish between their two workgroups as in mi_mla_decode_set_pair -- same sums in the same
 * order as the merge kernel, bit-identical outputs; the meeting words live in the workspace mi_gqa_decode_workspace() sizes.  mode 0 / 1 =
 * off / on for the calls that follow, -1 = default (on; MI_GQA_PAIR=0 in the environment turns it off), 2 = on with the second piece
 * withholding its word (a test of the bounded wait and the merge kernel's take-over).  Process-wide knob. */
int mi_gqa_decode_set_pair(int mode);
int mi_gqa_decode(const void *q, const void *k, const void *v, void *out, const int32_t *kv_seq_lens,
                  const int32_t *block_table, int batch, int q_heads, int kv_heads, int k_dim, int v_dim, int page_size,
                  int bt_stride, int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t k_stride_blk,
                  int64_t k_stride_row, int64_t k_stride_h, int64_t v_stride_blk, int64_t v_stride_row, int64_t v_stride_h,
                  int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, int num_splits, void *workspace,
                  size_t workspace_bytes, void *stream);
/* The same kernel with attention sinks, a sliding window and one block-table row per query row (attention/sinks_attention.py:7-286):
 * sinks [q_heads] (sinks_dtype MI_DTYPE_BF16 / F16 / F32; NULL = none) = a per-head logit, not scaled by sm_scale, that enters the softmax
 * maximum and denominator but has no value row; sliding_window = keys [len - window, len) (-1 = all); block_table_rows [batch] = the
 * block-table row of query row b (NULL = b) -- the extend form runs every new token as a row of its own with kv_seq_lens = its causal length. */
int mi_gqa_decode_sinks(const void *q, const void *k, const void *v, void *out, const int32_t *kv_seq_lens, const int32_t *block_table, int batch,
                        int q_heads, int kv_heads, int k_dim, int v_dim, int page_size, int bt_stride, int max_seq_len, int64_t q_stride_b,
                        int64_t q_stride_h, int64_t k_stride_blk, int64_t k_stride_row, int64_t k_stride_h, int64_t v_stride_blk,
                        int64_t v_stride_row, int64_t v_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype, int num_splits,
                        void *workspace, size_t workspace_bytes, const void *sinks, int sinks_dtype, int sliding_window,
                        const int32_t *block_table_rows, void *stream);

/* ---- SwiGLU + per-row INT8 quantisation (swiglu_quant.py:87-127) -------------------------------------------------
 * x [rows, cols] (cols = 2I; gate = x[:, :I], up = x[:, I:]); only the first `total` rows are processed, where total =
 * group_list[num_groups-1] (group_list_type 0, cumulative) or sum(group_list) (type 1, counts), read on the device.
 * need_quant: out int8 [rows, I] + scale f32 [rows] (scale = max|v|/127, q = clamp(floor(v/scale + 0.5), -128, 127));
 * else out [rows, I] in the input dtype.  do_limit clamps gate <= limit and up to [-limit, limit]. */
int mi_swiglu_quant(const void *x, const void *group_list, int group_list_is_i64, int num_groups, int group_list_type,
                    int rows, int cols, int need_quant, int do_limit, float limit, int dtype, void *out, float *scale,
                    void *stream);

/* ---- Add + RMSNorm (+bias) (+static INT8 quant) and the Gemma variant (add_rmsnorm_bias.py:83-147,194-232) ----------
 * y = input (+ residual) in the I/O dtype -> out2 (may be NULL when residual is NULL);
 * v = float(y) * rstd * weight (+ bias)            [gemma != 0: rstd = rsqrt(var + eps), v = float(y)*rstd*(weight + 1)]
 * out = v in the I/O dtype, or int8 saturate(rint(v * quant_scale + quant_offset)) when quant_scale/offset are given.
 * weight / bias / quant_* are [hidden] in the I/O dtype; hidden % 8 == 0, hidden <= 8192. */
int mi_add_rmsnorm_bias(const void *input, const void *residual, const void *weight, const void *bias, float eps,
                        const void *quant_scale, const void *quant_offset, int gemma, int rows, int hidden,
                        int64_t input_row_stride, int dtype, void *out, void *out2, void *stream);

/* ---- split QKV + per-head RMSNorm (+bias) + RoPE (split_qkv_rmsnorm_rope.py:374-438) ---------------------------------
 * qkv [rows, q_hidden + 2*kv_hidden]; sin / cos [rows, rope_dim]; q/k/v outputs contiguous.  has_norm == 0 skips the
 * norm (reference: eps is None); biases optional (both or neither); rope_dim <= head_dim (partial RoPE); neox != 0 =
 * rotate-half layout, else interleaved pairs.  head_dim power of two. */
int mi_split_qkv_rmsnorm_rope(const void *qkv, const void *sin, const void *cos, int rows, int q_hidden, int kv_hidden,
                              int head_dim, int rope_dim, int has_norm, float eps, const void *q_weight,
                              const void *k_weight, const void *q_bias, const void *k_bias, int neox, int dtype, void *q,
                              void *k, void *v, void *stream);
/* ---- split [q | gate] + K + V, Gemma RMSNorm (weight + 1) + neox RoPE (split_qkv_rmsnorm_rope.py:441-745) ----------------------
 * input [rows, 2 q_hidden + 2 kv_hidden]: q_hidden / head_dim pairs [q head | gate head], then K, then V; sin / cos [rows, rope_dim];
 * q, gate [rows, q_hidden], k, v [rows, kv_hidden]; head_dim a power of two in [8, 2048], q_hidden % kv_hidden == 0 (:700-702). */
int mi_split_qkvgate_gemma_rmsnorm_rope(const void *input, const void *sin, const void *cos, int rows, int q_hidden, int kv_hidden,
                                        int head_dim, int rope_dim, float eps, const void *q_weight, const void *k_weight, int dtype,
                                        void *q, void *k, void *v, void *gate, void *stream);

/* split QKV (+ gate) + per-head RMSNorm (+ bias) + multimodal RoPE (norm/split_qkv_rmsnorm_mrope.py:335-420): cos_sin [3, rows, rope_dim],
 * per section (t, h, w) the first half of a row = cos, the second = sin; rotation offset o < rope_dim / 2 reads section h when
 * (sections_interleaved: o % 3 == 1 and o <= 3 sec_h; else sec_t <= o < sec_t + sec_h), w when (o % 3 == 2 and o <= 3 sec_w; else
 * o >= sec_t + sec_h), otherwise t; rotate-half on the first rope_dim dims.  gate != NULL: the row starts with q_heads pairs
 * [q head | gate head] and the gates are copied out.  head_dim a power of two in [64, 256], rope_dim % 16 == 0. */
int mi_split_qkv_rmsnorm_mrope(const void *qkv, const void *cos_sin, int rows, int q_hidden, int kv_hidden, int head_dim, int rope_dim, float eps,
                               const void *q_weight, const void *k_weight, const void *q_bias, const void *k_bias, int sec_t, int sec_h,
                               int sec_w, int sections_interleaved, int dtype, void *q, void *k, void *v, void *gate, void *stream);

/* split QKV + optional per-head RMSNorm (+ bias) + rotate-half RoPE whose cos / sin come from a position-indexed cache
 * (norm/split_qkv_rmsnorm_rope_pos_cache_half_npu.py:232-407): cos_sin_cache [max_seq, cache_stride0 >= rope_dim] of cache_dtype (BF16 / F16 /
 * F32; the reference test passes fp32), row = [rope_dim / 2 cos | rope_dim / 2 sin]; positions [rows] int32 / int64, clamped to [0, max_seq).
 * cast_norm: the normalised value is rounded to the I/O dtype before the rotation (the reference's default).  has_norm = 0: no norm. */
int mi_split_qkv_rmsnorm_rope_pos_cache(const void *qkv, const void *positions, int pos_is_i64, const void *cos_sin_cache, int cache_dtype,
                                        int max_seq, long long cache_stride0, int rows, int q_hidden, int kv_hidden, int head_dim, int rope_dim,
                                        int has_norm, float eps, const void *q_weight, const void *k_weight, const void *q_bias, const void *k_bias,
                                        int cast_norm, int dtype, void *q, void *k, void *v, void *stream);

/* ---- row statistics and scalings (norm/l1_norm.py:7-38, norm/rmsnorm_without_weight.py:30-76, norm/rmsnorm_split.py:34-161) -------------
 * x [rows, cols] contiguous, dtype MI_DTYPE_BF16 / F16 / F32 (the reference tests use fp32); arithmetic in fp32.
 *   mi_l1_norm                 out fp32 [rows, cols] = x / sum(x) per row
 *   mi_rmsnorm_without_weight  out (dtype of x)      = x * rsqrt(sum(x^2) * (1 / cols) + eps)
 *   mi_row_variance            out (dtype of x) [rows] = sum(x^2) / cols
 *   mi_rsqrt_mul               out (dtype of x)      = (x * rsqrt(variance[row] + eps)) * weight[col]; variance [rows], weight [cols] in x's dtype */
int mi_l1_norm(const void *x, long long rows, int cols, int dtype, float *out, void *stream);
int mi_rmsnorm_without_weight(const void *x, long long rows, int cols, float eps, int dtype, void *out, void *stream);
int mi_row_variance(const void *x, long long rows, int cols, int dtype, void *out, void *stream);
int mi_rsqrt_mul(const void *x, const void *variance, const void *weight, long long rows, int cols, float eps, int dtype, void *out,
                 void *stream);
/* out = x * (c + scale) + shift (norm/scale_shift.py:122-183): x, out [rows, cols] in `dtype`; scale (1 or cols values) and shift (1, cols or
 * rows * cols values) in ss_dtype = dtype or MI_DTYPE_F32.  c = scale_constant when shift has one value per element (then scale must have
 * one per column), 1.0 otherwise -- the reference's two kernels (:60, :112). */
/* split QKV + tensor-parallel RMSNorm + RoPE (norm/split_qkv_tp_rmsnorm_rope.py:179-288) in two launches around the caller's all-reduce of
 * qk_var: (1) V copied, qk_var [rows, 2] fp32 = mean(q^2), mean(k^2) over this rank's columns; (2) q, k = rope(dtype((x * rsqrt-like scale) *
 * weight[col])) with scale = 1 / sqrt(qk_var * inv_tp_world + eps), neox rotation of the first rotary_dim dims of every head with the first
 * half of the row's cos / sin [rows, rotary_dim].  input [rows, q_cols + 2 k_cols]; weights [q_cols], [k_cols] in the I/O dtype. */
int mi_split_qkv_tp_var(const void *input, long long rows, int q_cols, int k_cols, int dtype, void *v, float *qk_var, void *stream);
int mi_split_qkv_tp_norm_rope(const void *input, const void *cos, const void *sin, const float *qk_var, long long rows, int q_cols, int k_cols,
                              int head_dim, int rotary_dim, float eps, float inv_tp_world, const void *q_weight, const void *k_weight, int dtype,
                              void *q, void *k, void *stream);
/* MLA down-projection row [q_lora_rank | kv_lora_rank | qk_rope_dim] -> RMSNorm(q part) * q_weight (+ q_bias), RMSNorm(kv part) * k_weight
 * (+ k_bias), rope part copied (norm/fused_split_qk_norm.py:93-134); weights / biases (nullable) in the I/O dtype; fp32 arithmetic. */
int mi_fused_split_qk_norm(const void *x, long long rows, int q_lora_rank, int kv_lora_rank, int qk_rope_dim, float eps, const void *q_weight,
                           const void *q_bias, const void *k_weight, const void *k_bias, int dtype, void *q_lora, void *k_nope, void *k_pe,
                           void *stream);
/* GPT-OSS SwiGLU (activation/swiglu_oai.py:53-83): x [rows, dim] with gate / up interleaved (even / odd columns) -> out [rows, dim / 2] =
 * (min(max(up, -limit), limit) + 1) * g * sigmoid(g * alpha), g = min(gate, limit); dim / 2 a multiple of 8 (4 for fp32). */
int mi_swiglu_oai(const void *x, long long rows, int dim, float alpha, float limit, int dtype, void *out, void *stream);
/* GPT-OSS SwiGLU on [gate | up] halves with optional per-row INT8 (activation/swiglu_oai_quant.py:115-211): x [rows, cols]; group_list NULL
 * (all rows) or num_groups counts / cumulative counts (type 1 / 0) bounding the rows that are computed; need_quant: out int8 [rows, cols / 2] +
 * scale fp32 [rows] (= max|out| / 127; q = trunc(dtype(out / scale)) saturated -- see rownorm.hip for the rounding assumption), else out in
 * x's dtype. */
int mi_swiglu_oai_quant(const void *x, const void *group_list, int group_list_is_i64, int num_groups, int group_list_type, long long rows, int cols,
                        float alpha, float limit, int need_quant, int dtype, void *out, float *scale, void *stream);
/* SiTU (activation/situ.py:11-480): x [rows, cols] = [gate | up]; out = beta * tanh(gate / beta) * sigmoid(gate) * up', up' = linear_beta *
 * tanh(up / linear_beta) (linear_beta <= 0: up' = up); group_list as in mi_swiglu_oai_quant; need_quant: int8 out + scale = max(max|out| / 127,
 * 1e-30), q = clamp(floor(out / scale + 0.5), -128, 127); else out in x's dtype. */
int mi_situ_and_mul(const void *x, const void *group_list, int group_list_is_i64, int num_groups, int group_list_type, long long rows, int cols,
                    float beta, float linear_beta, int need_quant, int dtype, void *out, float *scale, void *stream);
/* Kimi-K3 attention residual (kimi_k3/attn_residual.py:7-111): per token the num_valid_blocks bank rows and the prefix row are scored
 * (sum(rmsnorm(row) * combined_weight)), soft-maxed and mixed: out = sum_r p_r row_r.  prefix_sum [tokens, hidden], bank [tokens, blocks,
 * hidden], out [tokens, hidden] (row strides in elements, hidden contiguous; hidden % 8 == 0; num_valid_blocks <= 63);
 * combined_weight [hidden] in the I/O dtype or fp32. */
int mi_attn_residual_mix(const void *prefix_sum, long long stride_pm, const void *bank, long long stride_bm, long long stride_bb,
                         const void *combined_weight, int weight_dtype, long long tokens, int num_valid_blocks, int hidden, float eps, int dtype,
                         void *out, long long stride_om, void *stream);
/* out = routed * factor + shared (moe/mul_add.py:9-60), the product rounded to the I/O dtype before the sum. */
int mi_mul_add(const void *routed, const void *shared, float factor, long long numel, int dtype, void *out, void *stream);
/* "Zero experts" of type identity (moe/zero_experts_compute_identity.py:6-81): result [tokens, D] = hidden * sum of the scales of the
 * selections with index >= num_experts; IN PLACE those scales become 0 and those indices identity_mask_value (the first one 0 when all K
 * selections of the token were zero experts).  expert_indices [tokens, K] int32 / int64, expert_scales [tokens, K] fp32 or the I/O dtype. */
int mi_zero_experts_identity(void *expert_indices, int idx_is_i64, void *expert_scales, int scales_dtype, int num_experts, const void *hidden,
                             long long tokens, int K, int D, int identity_mask_value, int dtype, void *result, void *stream);
int mi_scale_shift(const void *x, const void *scale, const void *shift, long long rows, int cols, long long scale_numel, long long shift_numel,
                   float scale_constant, int dtype, int ss_dtype, void *out, void *stream);

/* ---- RoPE on q and the shared key heads (norm/fused_rope_qk_mqa.py:113-160) -----------------------------------------
 * q [tokens, q_heads, head_dim], k [tokens, k_heads, head_dim] (strides in elements, last dim contiguous); cos_sin
 * [tokens, rope_dim] = cos | sin halves, one row per token; first rope_dim dims rotated (neox != 0: pairs (i, i+rope/2),
 * else (2i, 2i+1)), the rest copied; every product and sum rounded to the I/O dtype like the reference kernel.
 * out_q / out_k contiguous. */
int mi_rope_qk_mqa(const void *q, const void *k, const void *cos_sin, int tokens, int q_heads, int k_heads, int head_dim,
                   int rope_dim, int neox, int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h,
                   int64_t cs_stride_t, int dtype, void *out_q, void *out_k, void *stream);

/* Per-query block tables of the sparse + causal prefill path (attention/fia_blockq_attention.py:11-88): topk_idx [total_q, topk1] int32,
 * -1-padded logical blocks; seq_lens [total_q] = position + 1; per_query_req [total_q] int32 / int64; req_to_token [requests, max_cols]
 * slot table (strides in elements).  Out: block_table [total_q, topk1] physical pages, the query's own block last, pads 0;
 * actual_kvlen [total_q].  The attention is mi_gqa_decode with one query row per "sequence" on these tables. */
int mi_fia_prep(const int32_t *topk_idx, long long stride_ti_t, long long stride_ti_k, const int32_t *seq_lens, const void *per_query_req,
                int req_is_i64, const int32_t *req_to_token, long long stride_rtt_r, long long stride_rtt_t, int max_cols, int total_q, int topk1,
                int block_size, int32_t *block_table, long long stride_bt_t, int32_t *actual_kvlen, void *stream);

/* ---- mla_preprocess (reference: one AscendC MIX kernel, csrc/mla_preprocess/op_kernel/mla_preprocess_mix_bf16.hpp:285,2762,2814;
 * host csrc/mla_preprocess/op_host/mla_preprocess.cpp:623-704; arithmetic per the test golden golden2_pytorch,
 * tests/python/sgl_kernel_npu/test_mla_preprocess.py:407-483).  Five launches on one stream, every GEMM hand-written:
 *   pre_quant -> pre_gemm_i8(mode 0) -> pre_mid -> pre_gemm_i8(mode 1) -> pre_bmm_rope
 * pre_quant:   out int8 = round(clamp(fp16(x / scale + zero_point), -128, 127)), numel % 8 == 0.
 * pre_gemm_i8: C[tokens, n] = A[tokens, k] int8 x W[n, k]^T int8 (W row-major: output channel major, K contiguous; k % 64 == 0).
 *              mode 0: split-K, one workgroup per 512-byte K-chunk: c_i32 [ceil(k/512)][tokens][n] partial products whose sum
 *                      (exact, order-independent) is the GEMM -- mi_mla_pre_gemm_i8_partials(k) of them;
 *              mode 1: y[tokens, n] (I/O dtype) = (float(c + bias[n])) * descale[n], one rounding (golden :95-107); bias may be NULL.
 * quant_mode "per_token_quant_symm" (the reference's default; mla_preprocess_mix_bf16.hpp:389-483,2262-2279): pre_quant_token writes
 *              int8 rows quantised against their own maximum (scale = max|x| / 127, q = rint(clamp(fp16(x * (1 / scale))))) and the
 *              per-token scales; pre_mid with tok_scale_in / tok_scale_out dequantises GEMM1 with (float(c) * descale0[j]) *
 *              tok_scale_in[t] (no bias) and requantises the normalised q against its own row maximum; pre_gemm_i8 mode 1 with
 *              row_scale multiplies by the token scale after the channel scale (pass bias = NULL).
 * pre_mid:     sum of the num_partials slices of gemm1_i32 [num_partials][tokens, 2112] (+bias0) * descale0 -> I/O dtype -> [512 k_nope | 64 k_pe | 1536 q];
 *              kv_cache[slot, :512] = rms_norm(k_nope) * gamma2, kv_cache_rope[slot, :64] = rope_half(k_pe, cos, sin),
 *              q_int8 [tokens, 1536] = per-tensor quant of rms_norm(q) * gamma1 + beta1.  cos / sin [tokens, 64].
 * pre_bmm_rope: y [tokens, q_heads*192] per head [128 nope | 64 pe]: q_out0[t, h, :512] = y_nope[t, h, :] x wuk_t[h]^T with
 *              wuk_t [q_heads, 512, 128] (the caller's wuk [q_heads, 128, 512] transposed once: K contiguous), fp32 accumulate;
 *              q_out1[t, h, :64] = rope_half(y_pe[t, h, :], cos, sin). */
int mi_mla_pre_quant(const void *x, const void *scale /*[1], I/O dtype*/, const int8_t *zero_point /*[1]*/, int64_t numel, int dtype,
                     int8_t *out, void *stream);
int mi_mla_pre_gemm_i8_partials(int k);
int mi_mla_pre_quant_token(const void *x, int tokens, int hidden, int dtype, int8_t *out, float *tok_scale, void *stream);
int mi_mla_pre_gemm_i8(const int8_t *a, int tokens, int k, const int8_t *w, int n, int mode, int32_t *c_i32, const int32_t *bias,
                       const float *descale, const float *row_scale /* NULL, or [tokens]: per-token dequant scale */, void *y,
                       int dtype, void *stream);
int mi_mla_pre_mid(const int32_t *gemm1_i32, int num_partials, const int32_t *bias0, const float *descale0, const void *gamma1, const void *beta1,
                   const void *gamma2, const void *cos, const void *sin, const int32_t *slotmapping, const void *quant_scale1,
                   const int8_t *quant_offset1, float eps, int tokens, int dtype, int8_t *q_int8, void *kv_cache, void *kv_cache_rope,
                   const float *tok_scale_in /* NULL = per-tensor mode */, float *tok_scale_out /* [tokens], with tok_scale_in */,
                   int cache_mode /* 1 krope_ctkv, 2 int8_nzcache, 3 nzcache (op_host/mla_preprocess.cpp:605-606) */,
                   int block_size /* slots per cache block (modes 2, 3) */, const void *ctkv_scale /* [1], I/O dtype (mode 2) */,
                   void *stream);
/* Cache layouts: mode 1 [slot][dim]; mode 3 per block of block_size slots [dim / 16][slot in block][16]; mode 2 k_nope as int8
 * = round(clamp(fp16(k_nope / ctkv_scale))) in [dim / 32][slot in block][32], k_pe as in mode 3 (element positions as read back by
 * the reference test's extract_from_nzcache, tests/python/sgl_kernel_npu/test_mla_preprocess.py:122-136).
 * q_nope_scale != NULL ([q_heads], I/O dtype; mode 2): q_out0 is int8 [tokens, q_heads, 512] = round(clamp(fp16(q * scale[h]))). */
int mi_mla_pre_bmm_rope(const void *y, int tokens, int q_heads, const void *wuk_t, const void *cos, const void *sin, int dtype,
                        void *q_out0, void *q_out1, const void *q_nope_scale, void *stream);
/* The WHOLE op in one launch (decode sizes): quantisation, GEMM1, the middle stage and the per-head stage run back to back inside one
 * grid, separated by three grid barriers; stage outputs travel through the same caller-provided buffers as with the separate calls
 * (a8 [tokens, hidden] int8, c1 [partials, tokens, 2112] int32, q8 [tokens, 1536] int8, tok0 / tok1 [tokens] float in per-token
 * mode) and the results are bit-identical to them.  sync_words: mi_mla_preprocess_one_launch_sync_words() uint32 words of device
 * memory, zero-initialised ONCE by the caller and lent to one call in flight at a time (the call counter that tags the barrier flags
 * lives in them: nothing the host passes changes from call to call, so the launch can be captured in a graph and replayed).  Returns MI_SGL_ENOTAPPLICABLE when the grid (q_heads x token blocks) exceeds the
 * number of CUs (all workgroups must be resident at once): issue the separate launches then.  A workgroup that is not joined by the
 * others within 2 s traps (the launch fails loudly instead of hanging or returning partial results). */
size_t mi_mla_preprocess_one_launch_sync_words(void);
int mi_mla_preprocess_one_launch(const void *hidden, int tokens, int hidden_size, const void *quant_scale0, const int8_t *quant_offset0,
                                 int8_t *a8, float *tok0, const int8_t *wdqkv, int32_t *c1, const int32_t *bias0, const float *descale0,
                                 const void *gamma1, const void *beta1, const void *gamma2, const void *cos, const void *sin,
                                 const int32_t *slotmapping, const void *quant_scale1, const int8_t *quant_offset1, float eps, int8_t *q8,
                                 void *kv_cache, void *kv_cache_rope, float *tok1, int cache_mode, int block_size, const void *ctkv_scale,
                                 const int8_t *wuq, int q_heads, const int32_t *bias1, const float *descale1, const void *wuk_t,
                                 void *q_out0, void *q_out1, const void *q_nope_scale, int per_token, int dtype, uint32_t *sync_words,
                                 void *stream);
/* GEMM2 + per-head BMM + RoPE in one launch (what the op runs): bit-identical to mi_mla_pre_gemm_i8(mode 1, k = 1536) followed by
 * mi_mla_pre_bmm_rope; the GEMM2 output y never goes to global memory.  a [tokens, 1536] int8, wuq [q_heads*192, 1536] int8. */
int mi_mla_pre_gemm2_bmm_rope(const int8_t *a, int tokens, const int8_t *wuq, int q_heads, const int32_t *bias, const float *descale,
                              const float *row_scale, const void *wuk_t, const void *cos, const void *sin, int dtype, void *q_out0,
                              void *q_out1, const void *q_nope_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_SGL_KERNELS_H_ */
