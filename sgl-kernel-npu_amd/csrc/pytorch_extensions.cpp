// torch.ops.npu.* registry for MI355X.
// Mirrors the reference registration unit csrc/pytorch_extensions.cpp (schemas :24-201 under TORCH_LIBRARY_FRAGMENT(npu, m),
// implementations :205-316) for the hot-path operators: the namespace stays `npu` so SGLang's call sites
// (`torch.ops.npu.<op>`) are unchanged; the dispatch key is CUDA (= HIP on ROCm) instead of PrivateUse1.
// The implementations are thin host functions (namespace sglang::npu_kernel, like include/sgl_kenel_npu_ops.h:14-239)
// that validate arguments, allocate outputs and call the C-ABI of include/mi_sgl_kernels.h on the current stream.
#include <cstdlib>
#include <map>
#include <mutex>

#include <hip/hip_runtime_api.h>
#include <ATen/ATen.h>
#include <c10/util/string_view.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "mi_sgl_kernels.h"

namespace sglang {
namespace npu_kernel {

static void *cur_stream() { return (void *)c10::hip::getCurrentHIPStream().stream(); }

static int dtype_code(const at::Tensor &t)
{
    TORCH_CHECK(t.scalar_type() == at::kBFloat16 || t.scalar_type() == at::kHalf, "expected a bfloat16 / float16 tensor, got ",
                t.scalar_type());
    return t.scalar_type() == at::kBFloat16 ? MI_DTYPE_BF16 : MI_DTYPE_F16;
}

static int dtype_code3(const at::Tensor &t)        // the row statistics / scalings also take fp32 (the reference tests run them in fp32)
{
    if (t.scalar_type() == at::kFloat) return MI_DTYPE_F32;
    return dtype_code(t);
}

std::string sgl_kernel_npu_version() { return std::string("sgl-kernel-npu_amd 0.1 (") + mi_sgl_kernels_version() + ")"; }

at::Tensor decode_mla_plan(const at::Tensor &kv_seq_lens, int64_t num_kv_heads);
void decode_mla_planned(const at::Tensor &q, const at::Tensor &k_nope_buffer, const at::Tensor &k_rope_buffer, at::Tensor &att_out,
                        const at::Tensor &kv_seq_lens, double sm_scale, int64_t page_size, const at::Tensor &block_table, const at::Tensor &plan);

// The attention layers of a decode step call decode_mla with the SAME kv_seq_lens tensor: the work list of the planned form is built by
// the first of them and reused by the others (the reference signature has no plan argument, so a drop-in caller gets "plan once, run
// many" this way).  One entry per device: a WEAK reference to the kv_seq_lens tensor's storage (the entry pins neither the tensor nor its
// allocator block: when the caller drops the tensor -- a model reload -- the reference expires and the next call rebuilds; while it is
// alive the bytes cannot have been recycled for another tensor; views of one buffer -- a padded kv_seq_lens cut to the batch -- share it), its data pointer and version counter (an in-place write through torch rebuilds
// the list), kv head count and stream.  Writes that bypass the version counter (a raw-pointer kernel, a graph replay) leave a stale list
// in place -- which the kernels tolerate by construction (pieces are clamped to the current lengths, the last piece runs to their end):
// stale costs balance, never correctness.  Inference tensors carry no version counter and are not cached.  MI_MLA_PLAN_CACHE=0 turns the
// reuse off; num_splits = -1 asks for the planned form with a list of its own; torch.ops.npu.clear_mla_plan_cache() drops every entry
// (the only thing an entry owns is its list: batch * kv_heads * 8 B + 1 KB of device memory).
struct MlaPlanCacheEntry {
    c10::weak_intrusive_ptr<c10::StorageImpl> lens{c10::intrusive_ptr<c10::StorageImpl>()};
    const void *lens_ptr = nullptr;
    int64_t lens_numel = 0;
    at::Tensor plan;
    uint32_t version = 0;
    int64_t kv_heads = 0;
    void *stream = nullptr;
    unsigned long long capture = 0;      // id of the stream capture the list was built in (0: built eagerly)
};
static std::mutex g_plan_mu;
static std::map<int, MlaPlanCacheEntry> &plan_entries()
{
    static auto &entries = *new std::map<int, MlaPlanCacheEntry>();      // (never destroyed: tensors must not outlive the HIP context at exit)
    return entries;
}
void clear_mla_plan_cache()
{
    std::lock_guard<std::mutex> lk(g_plan_mu);
    plan_entries().clear();
}
static at::Tensor cached_mla_plan(const at::Tensor &kv_seq_lens, int64_t kv_heads)
{
    void *st = cur_stream();
    const uint32_t ver = kv_seq_lens._version();
    // a list built while a graph is being captured has no contents until the graph runs: it serves the calls of the SAME capture only (the
    // layers of the captured step), never an eager call or another capture
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    unsigned long long cap = 0;
    if (hipStreamGetCaptureInfo((hipStream_t)st, &cap_status, &cap) != hipSuccess) {
        (void)hipGetLastError();         // a failed query must not sit in the thread's last-error slot: the plan launch checks it right after
        cap = 0;
    } else if (cap_status != hipStreamCaptureStatusActive) {
        cap = 0;
    }
    std::lock_guard<std::mutex> lk(g_plan_mu);
    MlaPlanCacheEntry &e = plan_entries()[kv_seq_lens.device().index()];
    // same storage, still alive (expired once the caller dropped every tensor on it), same bytes, unmodified through torch
    const bool alive = e.lens_ptr != nullptr && e.lens._unsafe_get_target() == kv_seq_lens.storage().unsafeGetStorageImpl() && !e.lens.expired();
    if (alive && e.lens_ptr == kv_seq_lens.data_ptr() && e.lens_numel == kv_seq_lens.numel() && e.version == ver &&
        e.kv_heads == kv_heads && e.stream == st && e.capture == cap)
        return e.plan;
    e.plan = decode_mla_plan(kv_seq_lens, kv_heads);
    e.lens = kv_seq_lens.storage().getWeakStorageImpl();
    e.lens_ptr = kv_seq_lens.data_ptr(), e.lens_numel = kv_seq_lens.numel();
    e.version = ver, e.kv_heads = kv_heads, e.stream = st, e.capture = cap;
    return e.plan;
}

// Paged MLA decode, same argument meaning as the reference Python entry point decode_mla
// (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:166-175); writes att_out in place.
void decode_mla(const at::Tensor &q, const at::Tensor &k_nope_buffer, const at::Tensor &k_rope_buffer, at::Tensor &att_out,
                const at::Tensor &kv_seq_lens, double sm_scale, int64_t page_size, const at::Tensor &block_table,
                int64_t num_splits)
{
    const c10::DeviceGuard device_guard(q.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(q.dim() == 3 && k_nope_buffer.dim() == 4 && k_rope_buffer.dim() == 4 && att_out.dim() == 3 && block_table.dim() == 2,
                "decode_mla: bad ranks");
    TORCH_CHECK(k_nope_buffer.size(3) == 512 && k_rope_buffer.size(3) == 64 && q.size(2) == 576 && att_out.size(2) == 512,
                "decode_mla: this build supports the MLA head layout 512 (nope) + 64 (rope)");
    TORCH_CHECK(q.stride(2) == 1 && k_nope_buffer.stride(3) == 1 && k_rope_buffer.stride(3) == 1 && att_out.stride(2) == 1,
                "decode_mla: innermost dimension must be contiguous");
    TORCH_CHECK(k_nope_buffer.size(1) == page_size && k_rope_buffer.size(1) == page_size, "decode_mla: page_size mismatch");
    TORCH_CHECK(q.scalar_type() == k_nope_buffer.scalar_type() && q.scalar_type() == k_rope_buffer.scalar_type() &&
                    q.scalar_type() == att_out.scalar_type(), "decode_mla: dtype mismatch");
    TORCH_CHECK(kv_seq_lens.scalar_type() == at::kInt && block_table.scalar_type() == at::kInt && kv_seq_lens.is_contiguous(),
                "decode_mla: kv_seq_lens / block_table must be int32");
    TORCH_CHECK(block_table.stride(1) == 1, "decode_mla: block_table rows must be contiguous");
    const int B = (int)q.size(0), Hq = (int)q.size(1), Hkv = (int)k_nope_buffer.size(2);
    TORCH_CHECK(Hq % Hkv == 0 && k_rope_buffer.size(2) == Hkv, "decode_mla: head counts");
    // The batch is q's (as in the reference, decode_attention.py:178); a longer (padded, graph-static) kv_seq_lens is cut to it HERE: the
    // work list's item offsets depend on the (sequence, kv head) pair count, so list builder and consumers must agree on it.
    TORCH_CHECK(kv_seq_lens.dim() == 1 && kv_seq_lens.size(0) >= B, "decode_mla: kv_seq_lens must be [batch] with batch >= q.size(0)");
    if (kv_seq_lens.size(0) > B) {
        decode_mla(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens.narrow(0, 0, B), sm_scale, page_size, block_table, num_splits);
        return;
    }
    const int max_len = (int)std::min<int64_t>(block_table.size(1) * page_size, INT32_MAX);   // upper bound, no host sync
    int splits = (int)num_splits;
    if (splits == 0) {
        splits = mi_mla_decode_num_splits(B, Hq, Hkv, max_len);
        static const bool reuse = !(getenv("MI_MLA_PLAN_CACHE") && atoi(getenv("MI_MLA_PLAN_CACHE")) == 0);
        if (splits == MI_MLA_SPLITS_PLANNED && reuse && B > 0 && !kv_seq_lens.is_inference()) {
            decode_mla_planned(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, sm_scale, page_size, block_table, cached_mla_plan(kv_seq_lens, Hkv));
            return;
        }
    }
    const size_t wsb = mi_mla_decode_workspace(B, Hq, splits);
    at::Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 16)}, at::dtype(at::kByte).device(q.device()));
    const int rc = mi_mla_decode(q.data_ptr(), k_nope_buffer.data_ptr(), k_rope_buffer.data_ptr(), att_out.data_ptr(),
                                 kv_seq_lens.data_ptr<int>(), block_table.data_ptr<int>(), B, Hq, Hkv, (int)page_size,
                                 (int)block_table.stride(0), max_len, q.stride(0), q.stride(1), k_nope_buffer.stride(0),
                                 k_nope_buffer.stride(1), k_nope_buffer.stride(2), k_rope_buffer.stride(0),
                                 k_rope_buffer.stride(1), k_rope_buffer.stride(2), att_out.stride(0), att_out.stride(1),
                                 (float)sm_scale, dtype_code(q), splits, ws.data_ptr(), wsb, cur_stream());
    TORCH_CHECK(rc == 0, "mi_mla_decode failed with code ", rc);
}

// MI355X extension (plan once, run many): the length-aware work list of decode_mla depends only on kv_seq_lens -- the attention layers of a
// decode step share it.  decode_mla_plan builds it (one small launch, no host sync); decode_mla_planned is decode_mla with that list
// instead of a plan launch of its own.  Where the planned form does not serve the shape, the planned call runs plain decode_mla.
at::Tensor decode_mla_plan(const at::Tensor &kv_seq_lens, int64_t num_kv_heads)
{
    const c10::DeviceGuard device_guard(kv_seq_lens.device());
    TORCH_CHECK(kv_seq_lens.scalar_type() == at::kInt && kv_seq_lens.is_contiguous() && kv_seq_lens.dim() == 1, "decode_mla_plan: kv_seq_lens must be int32 [batch]");
    const int B = (int)kv_seq_lens.size(0);
    const size_t bytes = mi_mla_decode_plan_bytes(B, (int)num_kv_heads);
    at::Tensor plan = at::empty({(int64_t)std::max<size_t>(bytes / 4, 4)}, at::dtype(at::kInt).device(kv_seq_lens.device()));
    if (B > 0)
        TORCH_CHECK(0 == mi_mla_decode_build_plan(kv_seq_lens.data_ptr<int>(), B, (int)num_kv_heads, plan.data_ptr(), bytes, cur_stream()),
                    "mi_mla_decode_build_plan failed");
    return plan;
}
void decode_mla_planned(const at::Tensor &q, const at::Tensor &k_nope_buffer, const at::Tensor &k_rope_buffer, at::Tensor &att_out,
                        const at::Tensor &kv_seq_lens, double sm_scale, int64_t page_size, const at::Tensor &block_table, const at::Tensor &plan)
{
    const c10::DeviceGuard device_guard(q.device());
    TORCH_CHECK(q.dim() == 3 && k_nope_buffer.dim() == 4 && k_rope_buffer.dim() == 4 && att_out.dim() == 3 && block_table.dim() == 2, "decode_mla: bad ranks");
    const int B = (int)q.size(0), Hq = (int)q.size(1), Hkv = (int)k_nope_buffer.size(2);
    // exact size: the list's item offsets depend on batch * kv heads (the size is strictly increasing in it), so a list built for a larger
    // batch must be refused, not read with the wrong offsets
    TORCH_CHECK(plan.scalar_type() == at::kInt && plan.is_contiguous() &&
                    (size_t)plan.numel() * 4 == std::max<size_t>(mi_mla_decode_plan_bytes(B, Hkv), 16),
                "decode_mla_planned: plan does not belong to this batch / kv head count");
    TORCH_CHECK(kv_seq_lens.dim() == 1 && kv_seq_lens.size(0) == B, "decode_mla_planned: kv_seq_lens must be [q.size(0)]");
    const bool shapes_ok = k_nope_buffer.size(3) == 512 && k_rope_buffer.size(3) == 64 && q.size(2) == 576 && att_out.size(2) == 512 &&
                           q.stride(2) == 1 && k_nope_buffer.stride(3) == 1 && k_rope_buffer.stride(3) == 1 && att_out.stride(2) == 1 &&
                           k_nope_buffer.size(1) == page_size && k_rope_buffer.size(1) == page_size && kv_seq_lens.scalar_type() == at::kInt &&
                           block_table.scalar_type() == at::kInt && kv_seq_lens.is_contiguous() && block_table.stride(1) == 1 && Hq % Hkv == 0;
    if (shapes_ok && B > 0) {
        const int max_len = (int)std::min<int64_t>(block_table.size(1) * page_size, INT32_MAX);
        const size_t wsb = mi_mla_decode_workspace(B, Hq, MI_MLA_SPLITS_PLANNED);
        at::Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 16)}, at::dtype(at::kByte).device(q.device()));
        const int rc = mi_mla_decode_with_plan(q.data_ptr(), k_nope_buffer.data_ptr(), k_rope_buffer.data_ptr(), att_out.data_ptr(),
                                               kv_seq_lens.data_ptr<int>(), block_table.data_ptr<int>(), B, Hq, Hkv, (int)page_size,
                                               (int)block_table.stride(0), max_len, q.stride(0), q.stride(1), k_nope_buffer.stride(0),
                                               k_nope_buffer.stride(1), k_nope_buffer.stride(2), k_rope_buffer.stride(0), k_rope_buffer.stride(1),
                                               k_rope_buffer.stride(2), att_out.stride(0), att_out.stride(1), (float)sm_scale, dtype_code(q),
                                               plan.data_ptr(), ws.data_ptr(), wsb, cur_stream());
        if (rc == 0) return;
        TORCH_CHECK(rc == MI_SGL_ENOTAPPLICABLE, "mi_mla_decode_with_plan failed with code ", rc);
    }
    // all checks live there; -1: the planned form with a list of its own where it applies, uniform splits where it does not (never back here)
    decode_mla(q, k_nope_buffer, k_rope_buffer, att_out, kv_seq_lens, sm_scale, page_size, block_table, MI_MLA_SPLITS_PLANNED);
}

// Paged GQA decode with a separate V cache; argument meaning of decode_gqa (decode_attention.py:378-387); writes att_out in place.
// A DeepSeek-style cache where V is the first 512 columns of the 576-wide K rows (the reference special-cases Lk == 576,
// :404-407, and its test builds exactly that view, test_decode_attention.py:74) goes to the MLA kernel, which reads each K
// row once for both GEMMs; everything else runs the generic kernel (csrc/kernels/gqa_decode.hip).
void decode_gqa(const at::Tensor &q, const at::Tensor &k_buffer, const at::Tensor &v_buffer, at::Tensor &att_out,
                const at::Tensor &kv_seq_lens, double sm_scale, int64_t page_size, const at::Tensor &block_table, int64_t num_splits)
{
    const c10::DeviceGuard device_guard(q.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(q.dim() == 3 && k_buffer.dim() == 4 && v_buffer.dim() == 4 && att_out.dim() == 3 && block_table.dim() == 2,
                "decode_gqa: bad ranks");
    TORCH_CHECK(q.stride(2) == 1 && k_buffer.stride(3) == 1 && v_buffer.stride(3) == 1 && att_out.stride(2) == 1,
                "decode_gqa: innermost dimension must be contiguous");
    TORCH_CHECK(k_buffer.size(1) == page_size && v_buffer.size(1) == page_size, "decode_gqa: page_size mismatch");
    TORCH_CHECK(q.scalar_type() == k_buffer.scalar_type() && q.scalar_type() == v_buffer.scalar_type() &&
                    q.scalar_type() == att_out.scalar_type(), "decode_gqa: dtype mismatch");
    TORCH_CHECK(kv_seq_lens.scalar_type() == at::kInt && block_table.scalar_type() == at::kInt && kv_seq_lens.is_contiguous(),
                "decode_gqa: kv_seq_lens / block_table must be int32");
    TORCH_CHECK(block_table.stride(1) == 1, "decode_gqa: block_table rows must be contiguous");
    const int B = (int)q.size(0), Hq = (int)q.size(1), Hkv = (int)k_buffer.size(2);
    const int Lk = (int)k_buffer.size(3), Lv = (int)v_buffer.size(3);
    TORCH_CHECK(Hq % Hkv == 0, "head_num must be divisible by kv_head_num");
    TORCH_CHECK(v_buffer.size(2) == Hkv && q.size(2) == Lk && att_out.size(2) == Lv && att_out.size(0) == B && att_out.size(1) == Hq,
                "decode_gqa: shape mismatch");
    const int max_len = (int)std::min<int64_t>(block_table.size(1) * page_size, INT32_MAX);   // upper bound, no host sync
    int splits = (int)num_splits;
    const bool v_is_k_prefix = Lk == 576 && Lv == 512 && v_buffer.data_ptr() == k_buffer.data_ptr() &&
                               v_buffer.stride(0) == k_buffer.stride(0) && v_buffer.stride(1) == k_buffer.stride(1) &&
                               v_buffer.stride(2) == k_buffer.stride(2);
    if (v_is_k_prefix && splits == 0 && k_buffer.size(1) == page_size) {
        // the library's choice: exactly decode_mla on the two column ranges of the K rows (the layers of a step then share its work list too)
        decode_mla(q, k_buffer.narrow(3, 0, 512), k_buffer.narrow(3, 512, 64), att_out, kv_seq_lens, sm_scale, page_size, block_table, 0);
        return;
    }
    if (v_is_k_prefix) {
        if (splits == 0) splits = mi_mla_decode_num_splits(B, Hq, Hkv, max_len);
        const size_t wsb = mi_mla_decode_workspace(B, Hq, splits);
        at::Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 16)}, at::dtype(at::kByte).device(q.device()));
        const char *kb = (const char *)k_buffer.data_ptr();
        const int rc = mi_mla_decode(q.data_ptr(), kb, kb + 512 * 2, att_out.data_ptr(), kv_seq_lens.data_ptr<int>(),
                                     block_table.data_ptr<int>(), B, Hq, Hkv, (int)page_size, (int)block_table.stride(0), max_len,
                                     q.stride(0), q.stride(1), k_buffer.stride(0), k_buffer.stride(1), k_buffer.stride(2),
                                     k_buffer.stride(0), k_buffer.stride(1), k_buffer.stride(2), att_out.stride(0),
                                     att_out.stride(1), (float)sm_scale, dtype_code(q), splits, ws.data_ptr(), wsb, cur_stream());
        TORCH_CHECK(rc == 0, "mi_mla_decode failed with code ", rc);
        return;
    }
    if (splits == 0) splits = mi_gqa_decode_num_splits(B, Hq, Hkv, max_len);
    const size_t wsb = mi_gqa_decode_workspace(B, Hq, Lv, splits);
    at::Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 16)}, at::dtype(at::kByte).device(q.device()));
    const int rc = mi_gqa_decode(q.data_ptr(), k_buffer.data_ptr(), v_buffer.data_ptr(), att_out.data_ptr(), kv_seq_lens.data_ptr<int>(),
                                 block_table.data_ptr<int>(), B, Hq, Hkv, Lk, Lv, (int)page_size, (int)block_table.stride(0), max_len,
                                 q.stride(0), q.stride(1), k_buffer.stride(0), k_buffer.stride(1), k_buffer.stride(2),
                                 v_buffer.stride(0), v_buffer.stride(1), v_buffer.stride(2), att_out.stride(0), att_out.stride(1),
                                 (float)sm_scale, dtype_code(q), splits, ws.data_ptr(), wsb, cur_stream());
    TORCH_CHECK(rc == 0, "mi_gqa_decode failed with code ", rc,
                " (head dims must be multiples of 8 with (k, v) <= (64,64) (128,128) (192,128) (256,256) (288,256) or (576,512))");
}

// attention/sinks_attention.py:90-137 (decode) and :241-286 (extend): query [rows, Hq * D]; kv_lens [rows] int32 = keys each query row sees;
// bt_rows [rows] int32 = its block-table row (undefined tensor = row index).  Returns [rows, Hq * Dv].
at::Tensor attention_sinks(const at::Tensor &query, const at::Tensor &k_cache, const at::Tensor &v_cache, const at::Tensor &sinks,
                           const at::Tensor &block_tables, const at::Tensor &kv_lens, double scale, int64_t sliding_window_size, int64_t q_head_num,
                           int64_t k_head_num, const std::optional<at::Tensor> &bt_rows)
{
    const c10::DeviceGuard device_guard(query.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(query.dim() == 2 && query.is_contiguous() && k_cache.dim() == 4 && v_cache.dim() == 4 && block_tables.dim() == 2,
                "attention_sinks: query [rows, Hq * D], caches [blocks, page, Hkv, D], block_tables [seqs, max_blocks]");
    TORCH_CHECK(k_cache.stride(3) == 1 && v_cache.stride(3) == 1 && block_tables.stride(1) == 1, "attention_sinks: innermost dimensions must be contiguous");
    TORCH_CHECK(query.scalar_type() == k_cache.scalar_type() && query.scalar_type() == v_cache.scalar_type(), "attention_sinks: dtype mismatch");
    TORCH_CHECK(kv_lens.scalar_type() == at::kInt && kv_lens.is_contiguous() && block_tables.scalar_type() == at::kInt,
                "attention_sinks: kv lengths / block_tables must be int32");
    const int rows = (int)query.size(0), Hq = (int)q_head_num, Hkv = (int)k_head_num;
    TORCH_CHECK(Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && query.size(1) % Hq == 0 && k_cache.size(2) == Hkv && v_cache.size(2) == Hkv, "attention_sinks: heads");
    const int D = (int)(query.size(1) / Hq), Dv = (int)v_cache.size(3);
    TORCH_CHECK(k_cache.size(3) == D && kv_lens.numel() == rows, "attention_sinks: shape mismatch");
    TORCH_CHECK(sinks.numel() == Hq && sinks.is_contiguous(), "attention_sinks: sinks must hold one value per q head");
    if (bt_rows.has_value()) TORCH_CHECK(bt_rows->scalar_type() == at::kInt && bt_rows->is_contiguous() && bt_rows->numel() == rows, "attention_sinks: bt_rows");
    const int page = (int)k_cache.size(1);
    at::Tensor out = at::empty({rows, (int64_t)Hq * Dv}, query.options());
    int64_t max_len = std::min<int64_t>(block_tables.size(1) * (int64_t)page, INT32_MAX);
    if (sliding_window_size >= 0) max_len = std::min<int64_t>(max_len, std::max<int64_t>(sliding_window_size, 1));
    const int splits = mi_gqa_decode_num_splits(rows, Hq, Hkv, (int)max_len);
    const size_t wsb = mi_gqa_decode_workspace(rows, Hq, Dv, splits);
    at::Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 16)}, at::dtype(at::kByte).device(query.device()));
    const int rc = mi_gqa_decode_sinks(query.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), kv_lens.data_ptr<int>(),
                                       block_tables.data_ptr<int>(), rows, Hq, Hkv, D, Dv, page, (int)block_tables.stride(0), (int)max_len,
                                       (int64_t)Hq * D, D, k_cache.stride(0), k_cache.stride(1), k_cache.stride(2), v_cache.stride(0),
                                       v_cache.stride(1), v_cache.stride(2), (int64_t)Hq * Dv, Dv, (float)scale, dtype_code(query), splits,
                                       ws.data_ptr(), wsb, sinks.data_ptr(), dtype_code3(sinks), (int)sliding_window_size,
                                       bt_rows.has_value() ? bt_rows->data_ptr<int>() : nullptr, cur_stream());
    TORCH_CHECK(rc == 0, "mi_gqa_decode_sinks failed with code ", rc);
    return out;
}

// attention/fia_blockq_attention.py:89-181: per-query block tables (mi_fia_prep), then a paged decode with one query row per "sequence"
at::Tensor fia_blockq_sparse_prefill(const at::Tensor &q, const at::Tensor &k_cache, const at::Tensor &v_cache, const at::Tensor &topk_idx,
                                     const at::Tensor &seq_lens, const at::Tensor &per_query_req, const at::Tensor &req_to_token,
                                     int64_t block_size, double sm_scale, const std::optional<at::Tensor> &block_table_out,
                                     const std::optional<at::Tensor> &actual_kvlen_out)
{
    const c10::DeviceGuard device_guard(q.device());
    TORCH_CHECK(q.dim() == 3 && k_cache.dim() == 4 && topk_idx.dim() == 2 && req_to_token.dim() == 2, "fia_blockq_sparse_prefill: bad ranks");
    TORCH_CHECK(topk_idx.scalar_type() == at::kInt && seq_lens.scalar_type() == at::kInt && req_to_token.scalar_type() == at::kInt,
                "fia_blockq_sparse_prefill: topk_idx / seq_lens / req_to_token must be int32");
    TORCH_CHECK(per_query_req.scalar_type() == at::kInt || per_query_req.scalar_type() == at::kLong, "per_query_req must be int32 or int64");
    const int64_t total_q = q.size(0), topk1 = topk_idx.size(1);
    TORCH_CHECK(topk_idx.size(0) == total_q && seq_lens.numel() == total_q && per_query_req.numel() == total_q && seq_lens.is_contiguous() &&
                    per_query_req.is_contiguous() && topk1 <= 64 && k_cache.size(1) == block_size,
                "fia_blockq_sparse_prefill: shape mismatch (topk + 1 <= 64)");
    at::Tensor bt = block_table_out.has_value() ? *block_table_out : at::empty({total_q, topk1}, topk_idx.options());
    at::Tensor kvl = actual_kvlen_out.has_value() ? *actual_kvlen_out : at::empty({total_q}, topk_idx.options());
    TORCH_CHECK(bt.scalar_type() == at::kInt && bt.dim() == 2 && bt.size(0) == total_q && bt.size(1) == topk1 && bt.stride(1) == 1 &&
                    kvl.scalar_type() == at::kInt && kvl.numel() == total_q && kvl.is_contiguous(), "block_table_out / actual_kvlen_out");
    const int rc = mi_fia_prep(topk_idx.data_ptr<int>(), topk_idx.stride(0), topk_idx.stride(1), seq_lens.data_ptr<int>(), per_query_req.data_ptr(),
                               per_query_req.scalar_type() == at::kLong, req_to_token.data_ptr<int>(), req_to_token.stride(0),
                               req_to_token.stride(1), (int)req_to_token.size(1), (int)total_q, (int)topk1, (int)block_size, bt.data_ptr<int>(),
                               bt.stride(0), kvl.data_ptr<int>(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_fia_prep failed with code ", rc);
    at::Tensor out = at::empty({total_q, q.size(1), v_cache.size(3)}, q.options());
    if (total_q > 0) decode_gqa(q, k_cache, v_cache, out, kvl, sm_scale, block_size, bt, 0);
    return out;
}

// SwiGLU + per-row INT8 quantisation; same arguments / returns as swiglu_quant (activation/swiglu_quant.py:87-127).
std::tuple<at::Tensor, at::Tensor> swiglu_quant(const at::Tensor &x, const at::Tensor &group_list, int64_t group_list_type,
                                                bool need_quant, bool do_limit, double limit)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(group_list_type == 0 || group_list_type == 1, "group_list_type must be 0 or 1, but got ", group_list_type);
    TORCH_CHECK(x.dim() == 2 && x.is_contiguous(), "swiglu_quant: x must be a contiguous [s, h] tensor");
    TORCH_CHECK(group_list.scalar_type() == at::kInt || group_list.scalar_type() == at::kLong,
                "group_list dtype must be torch.int32 or torch.int64, but got ", group_list.scalar_type());
    TORCH_CHECK(group_list.dim() == 1 && group_list.is_contiguous(), "swiglu_quant: group_list must be 1-D contiguous");
    const int64_t s = x.size(0), h = x.size(1);
    at::Tensor out = at::empty({s, h / 2}, x.options().dtype(need_quant ? at::kChar : x.scalar_type()));
    at::Tensor scale = at::empty({s}, x.options().dtype(at::kFloat));
    const int rc = mi_swiglu_quant(x.data_ptr(), group_list.data_ptr(), group_list.scalar_type() == at::kLong,
                                   (int)group_list.size(0), (int)group_list_type, (int)s, (int)h, need_quant, do_limit,
                                   (float)limit, dtype_code(x), out.data_ptr(), scale.data_ptr<float>(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_swiglu_quant failed with code ", rc, " (h must be a multiple of 16 and <= 8192)");
    return {out, scale};
}

// Fused Add + RMSNorm (+bias) (+static INT8 quant); returns (output, residual_sum) like add_rmsnorm_bias
// (norm/add_rmsnorm_bias.py:83-147).  gemma = true gives add_gemma_rms_norm (:194-232) which returns (norm, add_output).
std::tuple<at::Tensor, at::Tensor> add_rmsnorm_bias(const at::Tensor &input, const std::optional<at::Tensor> &residual,
                                                    const at::Tensor &norm_weight, const std::optional<at::Tensor> &norm_bias,
                                                    double eps, const std::optional<at::Tensor> &quant_scale,
                                                    const std::optional<at::Tensor> &quant_offset, bool gemma)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.stride(1) == 1, "add_rmsnorm_bias: input must be [batch, hidden] with contiguous rows");
    const int64_t B = input.size(0), H = input.size(1);
    TORCH_CHECK(norm_weight.numel() == H && norm_weight.is_contiguous() && norm_weight.scalar_type() == input.scalar_type(),
                "add_rmsnorm_bias: weight must be [hidden] in the input dtype");
    TORCH_CHECK(quant_scale.has_value() == quant_offset.has_value(), "quant_scale and quant_offset go together");
    at::Tensor res_c;
    if (residual.has_value()) {
        TORCH_CHECK(residual->sizes() == input.sizes() && residual->scalar_type() == input.scalar_type(), "residual shape/dtype");
        res_c = (residual->stride(0) == input.stride(0) && residual->stride(1) == 1) ? *residual : residual->contiguous();
    }
    at::Tensor in_c = input;
    if (residual.has_value() && res_c.stride(0) != input.stride(0)) in_c = input.contiguous(), res_c = res_c.contiguous();
    at::Tensor out = at::empty({B, H}, input.options().dtype(quant_scale.has_value() ? at::kChar : input.scalar_type()));
    at::Tensor out2 = residual.has_value() ? at::empty({B, H}, input.options()) : input;
    auto opt_ptr = [&](const std::optional<at::Tensor> &t) -> const void * {
        if (!t.has_value()) return nullptr;
        TORCH_CHECK(t->numel() == H && t->is_contiguous() && t->scalar_type() == input.scalar_type(),
                    "add_rmsnorm_bias: per-column vectors must be [hidden] in the input dtype");
        return t->data_ptr();
    };
    const int rc = mi_add_rmsnorm_bias(in_c.data_ptr(), residual.has_value() ? res_c.data_ptr() : nullptr, norm_weight.data_ptr(),
                                       opt_ptr(norm_bias), (float)eps, opt_ptr(quant_scale), opt_ptr(quant_offset), gemma, (int)B,
                                       (int)H, in_c.stride(0), dtype_code(input), out.data_ptr(),
                                       residual.has_value() ? out2.data_ptr() : nullptr, cur_stream());
    TORCH_CHECK(rc == 0, "mi_add_rmsnorm_bias failed with code ", rc, " (hidden must be a multiple of 8 and <= 8192)");
    return {out, out2};
}

// RoPE on q and the shared key heads; arguments / returns of fused_rope_qk_mqa (norm/fused_rope_qk_mqa.py:113-160).
std::tuple<at::Tensor, at::Tensor> fused_rope_qk_mqa(const at::Tensor &query, const at::Tensor &key, const at::Tensor &cos_sin,
                                                     int64_t rotary_dim, bool is_neox_style)
{
    const c10::DeviceGuard device_guard(query.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(query.dim() == 3 && key.dim() == 3 && cos_sin.dim() == 2, "fused_rope_qk_mqa: query [T,Hq,D], key [T,Hk,D], cos_sin [T,R]");
    TORCH_CHECK(query.size(0) == key.size(0) && query.size(2) == key.size(2) && cos_sin.size(0) >= query.size(0) &&
                    cos_sin.size(1) >= rotary_dim, "fused_rope_qk_mqa: shape mismatch");
    TORCH_CHECK(query.stride(2) == 1 && key.stride(2) == 1 && cos_sin.stride(1) == 1, "fused_rope_qk_mqa: innermost dims must be contiguous");
    TORCH_CHECK(query.scalar_type() == key.scalar_type() && query.scalar_type() == cos_sin.scalar_type(), "fused_rope_qk_mqa: dtype mismatch");
    at::Tensor out_q = at::empty(query.sizes(), query.options()), out_k = at::empty(key.sizes(), key.options());
    const int rc = mi_rope_qk_mqa(query.data_ptr(), key.data_ptr(), cos_sin.data_ptr(), (int)query.size(0), (int)query.size(1),
                                  (int)key.size(1), (int)query.size(2), (int)rotary_dim, is_neox_style, query.stride(0), query.stride(1),
                                  key.stride(0), key.stride(1), cos_sin.stride(0), dtype_code(query), out_q.data_ptr(), out_k.data_ptr(),
                                  cur_stream());
    TORCH_CHECK(rc == 0, "mi_rope_qk_mqa failed with code ", rc);
    return {out_q, out_k};
}

// split QKV + per-head RMSNorm + RoPE; arguments as split_qkv_rmsnorm_rope (norm/split_qkv_rmsnorm_rope.py:374-438).
std::tuple<at::Tensor, at::Tensor, at::Tensor> split_qkv_rmsnorm_rope(
    const at::Tensor &input, const at::Tensor &sin, const at::Tensor &cos, int64_t q_hidden_size, int64_t kv_hidden_size,
    int64_t head_dim, std::optional<double> eps, const std::optional<at::Tensor> &q_weight,
    const std::optional<at::Tensor> &k_weight, const std::optional<at::Tensor> &q_bias, const std::optional<at::Tensor> &k_bias,
    bool is_neox_style)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.is_contiguous(), "split_qkv_rmsnorm_rope: input must be contiguous [batch, q+2kv]");
    TORCH_CHECK((head_dim & (head_dim - 1)) == 0, "head_dim must be a power of two");        // reference :390-391
    TORCH_CHECK(q_hidden_size % kv_hidden_size == 0, "q_hidden_size % kv_hidden_size != 0");   // reference :392
    TORCH_CHECK(input.size(1) == q_hidden_size + 2 * kv_hidden_size, "split_qkv_rmsnorm_rope: input width");
    const int64_t B = input.size(0);
    const int64_t rope_dim = sin.size(-1);
    TORCH_CHECK(sin.numel() == B * rope_dim && cos.numel() == B * rope_dim && sin.is_contiguous() && cos.is_contiguous() &&
                    sin.scalar_type() == input.scalar_type() && cos.scalar_type() == input.scalar_type(),
                "split_qkv_rmsnorm_rope: sin/cos must be contiguous [batch, ..., rope_dim] in the input dtype");
    if (eps.has_value()) TORCH_CHECK(q_weight.has_value() && k_weight.has_value(), "norm weights are required when eps is given");
    TORCH_CHECK(q_bias.has_value() == k_bias.has_value(), "q_bias and k_bias go together");
    at::Tensor q = at::empty({B, q_hidden_size}, input.options()), k = at::empty({B, kv_hidden_size}, input.options()),
               v = at::empty({B, kv_hidden_size}, input.options());
    auto p = [](const std::optional<at::Tensor> &t) -> const void * { return t.has_value() ? t->data_ptr() : nullptr; };
    const int rc = mi_split_qkv_rmsnorm_rope(input.data_ptr(), sin.data_ptr(), cos.data_ptr(), (int)B, (int)q_hidden_size,
                                             (int)kv_hidden_size, (int)head_dim, (int)rope_dim, eps.has_value(),
                                             (float)eps.value_or(0.0), p(q_weight), p(k_weight), p(q_bias), p(k_bias),
                                             is_neox_style, dtype_code(input), q.data_ptr(), k.data_ptr(), v.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_split_qkv_rmsnorm_rope failed with code ", rc);
    return {q, k, v};
}

// norm/l1_norm.py:29-38: fp32 [batch, hidden] = input / sum(input, -1)
at::Tensor l1_norm(const at::Tensor &input)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.is_contiguous(), "l1_norm: input must be contiguous [batch, hidden]");
    at::Tensor out = at::empty(input.sizes(), input.options().dtype(at::kFloat));
    const int rc = mi_l1_norm(input.data_ptr(), input.size(0), (int)input.size(1), dtype_code3(input), out.data_ptr<float>(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_l1_norm failed with code ", rc);
    return out;
}

// norm/rmsnorm_without_weight.py:59-76: x [B, L, C] -> x * rsqrt(mean(x^2, -1) + eps), same dtype
at::Tensor rmsnorm_without_weight(const at::Tensor &x, double eps)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() >= 1 && x.is_contiguous(), "fused_rmsnorm_without_weight: x must be contiguous");
    at::Tensor out = at::empty_like(x);
    const int64_t cols = x.size(-1);
    const int rc = mi_rmsnorm_without_weight(x.data_ptr(), cols ? x.numel() / cols : 0, (int)cols, (float)eps, dtype_code3(x), out.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_rmsnorm_without_weight failed with code ", rc);
    return out;
}

// norm/rmsnorm_split.py:150-161: x [B, L, C] -> mean(x^2, -1) as [B, L, 1] in x's dtype
at::Tensor fused_variance(const at::Tensor &x)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() == 3 && x.is_contiguous(), "fused_variance: x must be contiguous [B, L, C]");
    at::Tensor out = at::empty({x.size(0), x.size(1), 1}, x.options());
    const int rc = mi_row_variance(x.data_ptr(), x.size(0) * x.size(1), (int)x.size(2), dtype_code3(x), out.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_row_variance failed with code ", rc);
    return out;
}

// norm/rmsnorm_split.py:76-97: x [B, L, C], variance [B * L], weight [C] -> x * rsqrt(variance + eps) * weight in x's dtype
at::Tensor fused_rsqrt_mul(const at::Tensor &x, const at::Tensor &variance, const at::Tensor &weight, double eps)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() == 3 && x.is_contiguous(), "fused_rsqrt_mul: x must be contiguous [B, L, C]");
    const int64_t rows = x.size(0) * x.size(1), cols = x.size(2);
    TORCH_CHECK(variance.numel() == rows && variance.is_contiguous() && variance.scalar_type() == x.scalar_type(),
                "fused_rsqrt_mul: variance must hold B * L values in x's dtype");
    TORCH_CHECK(weight.numel() == cols && weight.is_contiguous() && weight.scalar_type() == x.scalar_type(),
                "fused_rsqrt_mul: weight must hold C values in x's dtype");
    at::Tensor out = at::empty_like(x);
    const int rc = mi_rsqrt_mul(x.data_ptr(), variance.data_ptr(), weight.data_ptr(), rows, (int)cols, (float)eps, dtype_code3(x), out.data_ptr(),
                                cur_stream());
    TORCH_CHECK(rc == 0, "mi_rsqrt_mul failed with code ", rc);
    return out;
}

// norm/scale_shift.py:122-183: x [B, L, C] * (c + scale) + shift
at::Tensor fused_scale_shift(const at::Tensor &x, const at::Tensor &scale, const at::Tensor &shift, double scale_constant)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() == 3 && x.is_contiguous(), "fused_scale_shift: x must be contiguous [B, L, C]");
    TORCH_CHECK(scale.is_contiguous() && shift.is_contiguous() && scale.scalar_type() == shift.scalar_type() &&
                    (scale.scalar_type() == x.scalar_type() || scale.scalar_type() == at::kFloat),
                "fused_scale_shift: scale and shift must be contiguous, of one dtype: x's or float32");
    const int64_t rows = x.size(0) * x.size(1), cols = x.size(2);
    TORCH_CHECK(scale.numel() == 1 || scale.numel() == cols, "scale must be scalar or [hidden_size]");                               // reference :136-138
    TORCH_CHECK(shift.numel() == 1 || shift.numel() == cols || shift.numel() == x.numel(), "shift must be scalar, [hidden_size] or x's size");      // :139-141
    at::Tensor out = at::empty_like(x);
    const int rc = mi_scale_shift(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), rows, (int)cols, scale.numel(), shift.numel(), (float)scale_constant,
                                  dtype_code3(x), dtype_code3(scale), out.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_scale_shift failed with code ", rc);
    return out;
}

// norm/split_qkv_rmsnorm_mrope.py:335-420 (triton_split_qkv_rmsnorm_mrope)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> split_qkv_rmsnorm_mrope(
    const at::Tensor &qkv, const at::Tensor &q_weight, const at::Tensor &k_weight, const at::Tensor &cos_sin, int64_t num_q_heads,
    int64_t num_kv_heads, int64_t head_size, double eps, at::IntArrayRef mrope_section, bool is_interleaved, std::optional<int64_t> rope_dim_opt,
    const std::optional<at::Tensor> &q_bias, const std::optional<at::Tensor> &k_bias, bool has_gate)
{
    const c10::DeviceGuard device_guard(qkv.device());   // launches and the current stream follow the tensor's GPU
    const int64_t q_size = num_q_heads * head_size, kv_size = num_kv_heads * head_size, gate_size = has_gate ? q_size : 0;
    const int64_t rope_dim = rope_dim_opt.value_or(head_size);
    TORCH_CHECK(qkv.dim() == 2 && qkv.is_contiguous() && qkv.size(1) == q_size + gate_size + 2 * kv_size,
                "split_qkv_rmsnorm_mrope: qkv must be contiguous [tokens, q (+ gate) + 2 kv]");
    const int64_t T = qkv.size(0);
    TORCH_CHECK(mrope_section.size() == 3, "mrope_section must hold three section sizes");
    TORCH_CHECK(cos_sin.dim() == 3 && cos_sin.size(0) == 3 && cos_sin.size(1) == T && cos_sin.size(2) == rope_dim && cos_sin.is_contiguous() &&
                    cos_sin.scalar_type() == qkv.scalar_type(),
                "split_qkv_rmsnorm_mrope: cos_sin must be contiguous [3, tokens, rope_dim] in the input dtype");
    TORCH_CHECK(q_weight.numel() == head_size && k_weight.numel() == head_size && q_weight.scalar_type() == qkv.scalar_type() &&
                    k_weight.scalar_type() == qkv.scalar_type() && q_weight.is_contiguous() && k_weight.is_contiguous(),
                "split_qkv_rmsnorm_mrope: weights must be [head_size] in the input dtype");
    TORCH_CHECK(q_bias.has_value() == k_bias.has_value(), "q_bias and k_bias go together");
    TORCH_CHECK((head_size & (head_size - 1)) == 0 && head_size >= 64 && head_size <= 256 && rope_dim % 16 == 0 && rope_dim <= head_size,
                "split_qkv_rmsnorm_mrope: head_size must be 64, 128 or 256 and rope_dim a multiple of 16 (this build)");
    at::Tensor q = at::empty({T, q_size}, qkv.options()), k = at::empty({T, kv_size}, qkv.options()), v = at::empty({T, kv_size}, qkv.options()),
               gate = at::empty({T, gate_size}, qkv.options());
    auto p = [](const std::optional<at::Tensor> &t) -> const void * { return t.has_value() ? t->data_ptr() : nullptr; };
    const int rc = mi_split_qkv_rmsnorm_mrope(qkv.data_ptr(), cos_sin.data_ptr(), (int)T, (int)q_size, (int)kv_size, (int)head_size, (int)rope_dim,
                                              (float)eps, q_weight.data_ptr(), k_weight.data_ptr(), p(q_bias), p(k_bias), (int)mrope_section[0],
                                              (int)mrope_section[1], (int)mrope_section[2], is_interleaved, dtype_code(qkv), q.data_ptr(),
                                              k.data_ptr(), v.data_ptr(), has_gate ? gate.data_ptr() : nullptr, cur_stream());
    TORCH_CHECK(rc == 0, "mi_split_qkv_rmsnorm_mrope failed with code ", rc);
    return {q, k, v, gate};
}

// activation/swiglu_oai.py:53-83 (swiglu_oai_triton)
at::Tensor swiglu_oai(const at::Tensor &hidden_states, int64_t dim, double gemm1_alpha, double gemm1_clamp_limit)
{
    const c10::DeviceGuard device_guard(hidden_states.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(hidden_states.is_contiguous() && dim > 0 && dim % 2 == 0 && hidden_states.numel() % dim == 0,
                "swiglu_oai: hidden_states must be contiguous with a multiple of dim elements");
    const int64_t rows = hidden_states.numel() / dim;
    at::Tensor out = at::empty({rows, dim / 2}, hidden_states.options());
    const int rc = mi_swiglu_oai(hidden_states.data_ptr(), rows, (int)dim, (float)gemm1_alpha, (float)gemm1_clamp_limit, dtype_code3(hidden_states),
                                 out.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_swiglu_oai failed with code ", rc, " (dim / 2 must be a multiple of 8)");
    return out;
}

// activation/swiglu_oai_quant.py:115-211
std::tuple<at::Tensor, at::Tensor> swiglu_oai_quant(const at::Tensor &x, double alpha, double limit, bool need_quant, const std::optional<at::Tensor> &group_list,
                                                    std::optional<int64_t> group_list_type)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() >= 1 && x.is_contiguous() && x.size(-1) % 2 == 0, "swiglu_oai_quant: x must be contiguous [..., 2d]");
    const int64_t h = x.size(-1), rows = h ? x.numel() / h : 0;
    if (group_list.has_value()) {
        TORCH_CHECK(group_list_type.has_value() && (*group_list_type == 0 || *group_list_type == 1), "group_list_type must be 0 or 1, got ",
                    group_list_type.value_or(-1));                                              // reference :151-152
        TORCH_CHECK(group_list->scalar_type() == at::kInt || group_list->scalar_type() == at::kLong, "group_list dtype must be torch.int32 or torch.int64");
        TORCH_CHECK(group_list->is_contiguous() && group_list->dim() == 1, "group_list must be a contiguous vector");
    }
    std::vector<int64_t> oshape(x.sizes().begin(), x.sizes().end());
    oshape.back() = h / 2;
    at::Tensor out = at::empty(oshape, x.options().dtype(need_quant ? at::kChar : x.scalar_type()));
    at::Tensor scale = at::empty({rows}, x.options().dtype(at::kFloat));
    const int rc = mi_swiglu_oai_quant(x.data_ptr(), group_list.has_value() ? group_list->data_ptr() : nullptr,
                                       group_list.has_value() && group_list->scalar_type() == at::kLong, group_list.has_value() ? (int)group_list->numel() : 0,
                                       (int)group_list_type.value_or(0), rows, (int)h, (float)alpha, (float)limit, need_quant, dtype_code(x), out.data_ptr(),
                                       scale.data_ptr<float>(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_swiglu_oai_quant failed with code ", rc);
    return {out, scale};
}

// activation/situ.py:165-480 (situ_and_mul, situ_and_mul_quant, situ share this op)
std::tuple<at::Tensor, at::Tensor> situ_and_mul(const at::Tensor &x, const std::optional<at::Tensor> &group_list, std::optional<int64_t> group_list_type,
                                                double beta, std::optional<double> linear_beta, bool need_quant)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() >= 1 && x.is_contiguous() && x.size(-1) % 2 == 0, "situ: x must be contiguous [..., 2d]");
    const int64_t h = x.size(-1), rows = h ? x.numel() / h : 0;
    if (group_list.has_value()) {
        TORCH_CHECK(group_list_type.has_value() && (*group_list_type == 0 || *group_list_type == 1), "group_list_type must be 0 or 1");
        TORCH_CHECK(group_list->scalar_type() == at::kInt || group_list->scalar_type() == at::kLong, "group_list dtype must be torch.int32 or torch.int64");
        TORCH_CHECK(group_list->is_contiguous() && group_list->dim() == 1, "group_list must be a contiguous vector");
    }
    std::vector<int64_t> oshape(x.sizes().begin(), x.sizes().end());
    oshape.back() = h / 2;
    at::Tensor out = at::empty(oshape, x.options().dtype(need_quant ? at::kChar : x.scalar_type()));
    at::Tensor scale = at::empty({rows}, x.options().dtype(at::kFloat));
    const int rc = mi_situ_and_mul(x.data_ptr(), group_list.has_value() ? group_list->data_ptr() : nullptr,
                                   group_list.has_value() && group_list->scalar_type() == at::kLong, group_list.has_value() ? (int)group_list->numel() : 0,
                                   (int)group_list_type.value_or(0), rows, (int)h, (float)beta, linear_beta.has_value() ? (float)*linear_beta : 0.f,
                                   need_quant, dtype_code(x), out.data_ptr(), scale.data_ptr<float>(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_situ_and_mul failed with code ", rc);
    return {out, scale};
}

// kimi_k3/attn_residual.py:66-111
at::Tensor attn_residual_mix(const at::Tensor &prefix_sum, const at::Tensor &bank, int64_t num_valid_blocks, const at::Tensor &combined_weight,
                             double variance_epsilon)
{
    const c10::DeviceGuard device_guard(prefix_sum.device());
    TORCH_CHECK(prefix_sum.dim() == 2 && bank.dim() == 3 && prefix_sum.stride(1) == 1 && bank.stride(2) == 1 && bank.size(0) == prefix_sum.size(0) &&
                    bank.size(2) == prefix_sum.size(1) && bank.scalar_type() == prefix_sum.scalar_type(),
                "mix_fused: prefix_sum [tokens, hidden], bank [tokens, blocks, hidden], hidden contiguous, one dtype");
    TORCH_CHECK(combined_weight.numel() == prefix_sum.size(1) && combined_weight.is_contiguous(), "mix_fused: combined_weight [hidden]");
    TORCH_CHECK(num_valid_blocks >= 0 && num_valid_blocks <= 63, "mix_fused: num_valid_blocks must be in [0, 63] in this build (got ", num_valid_blocks, ")");
    // the kernel moves 16-byte vectors: sliced views whose rows do not start on 16 bytes (bank[:, :, 4:], odd row strides) are copied first
    auto vec_ok = [](const at::Tensor &t, std::initializer_list<int64_t> strides) {
        if (reinterpret_cast<uintptr_t>(t.data_ptr()) % 16) return false;
        for (int64_t st : strides) if ((st * (int64_t)t.element_size()) % 16) return false;
        return true;
    };
    const at::Tensor prefix_sum_c = vec_ok(prefix_sum, {prefix_sum.stride(0)}) ? prefix_sum : prefix_sum.contiguous();
    const at::Tensor bank_c = vec_ok(bank, {bank.stride(0), bank.stride(1)}) ? bank : bank.contiguous();
    const at::Tensor &prefix_sum_ = prefix_sum_c, &bank_ = bank_c;
    at::Tensor out = at::empty_like(prefix_sum, at::MemoryFormat::Contiguous);
    const int rc = mi_attn_residual_mix(prefix_sum_.data_ptr(), prefix_sum_.stride(0), bank_.data_ptr(), bank_.stride(0), bank_.stride(1),
                                        combined_weight.data_ptr(), dtype_code3(combined_weight), prefix_sum.size(0), (int)num_valid_blocks,
                                        (int)prefix_sum.size(1), (float)variance_epsilon, dtype_code(prefix_sum), out.data_ptr(), out.stride(0),
                                        cur_stream());
    TORCH_CHECK(rc == 0, "mi_attn_residual_mix failed with code ", rc, " (hidden % 8 == 0; at most 63 blocks)");
    return out;
}

// moe/mul_add.py:38-60
at::Tensor mul_add(const at::Tensor &routed_input, const at::Tensor &shared_input, double scaling_factor)
{
    const c10::DeviceGuard device_guard(routed_input.device());
    TORCH_CHECK(routed_input.dim() == 2 && routed_input.is_contiguous() && shared_input.is_contiguous() && shared_input.sizes() == routed_input.sizes() &&
                    shared_input.scalar_type() == routed_input.scalar_type(), "mul_add: two contiguous [batch, hidden] tensors of one dtype");
    at::Tensor out = at::empty_like(routed_input);
    const int rc = mi_mul_add(routed_input.data_ptr(), shared_input.data_ptr(), (float)scaling_factor, routed_input.numel(), dtype_code3(routed_input),
                              out.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_mul_add failed with code ", rc);
    return out;
}

// moe/zero_experts_compute_identity.py:50-81 (expert_indices and expert_scales are modified in place)
at::Tensor zero_experts_compute_identity(at::Tensor &expert_indices, at::Tensor &expert_scales, int64_t num_experts, const at::Tensor &hidden_states,
                                         int64_t identity_mask_value)
{
    const c10::DeviceGuard device_guard(hidden_states.device());
    TORCH_CHECK(hidden_states.dim() == 2 && hidden_states.is_contiguous() && expert_indices.dim() == 2 && expert_indices.is_contiguous() &&
                    expert_scales.is_contiguous() && expert_scales.sizes() == expert_indices.sizes() && expert_indices.size(0) == hidden_states.size(0),
                "zero_experts_compute_identity: hidden [S, D], indices / scales [S, K], all contiguous");
    TORCH_CHECK(expert_indices.scalar_type() == at::kInt || expert_indices.scalar_type() == at::kLong, "expert_indices must be int32 or int64");
    TORCH_CHECK(expert_indices.size(1) <= 64, "zero_experts_compute_identity: K <= 64");
    at::Tensor out = at::empty_like(hidden_states);
    const int rc = mi_zero_experts_identity(expert_indices.data_ptr(), expert_indices.scalar_type() == at::kLong, expert_scales.data_ptr(),
                                            dtype_code3(expert_scales), (int)num_experts, hidden_states.data_ptr(), hidden_states.size(0),
                                            (int)expert_indices.size(1), (int)hidden_states.size(1), (int)identity_mask_value, dtype_code(hidden_states),
                                            out.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_zero_experts_identity failed with code ", rc);
    return out;
}

// norm/fused_split_qk_norm.py:93-134 (weights / biases of the two layer norms passed as tensors)
std::tuple<at::Tensor, at::Tensor, at::Tensor> fused_split_qk_norm(const at::Tensor &x, const at::Tensor &q_weight, const std::optional<at::Tensor> &q_bias,
                                                                   const at::Tensor &k_weight, const std::optional<at::Tensor> &k_bias, int64_t q_lora_rank,
                                                                   int64_t kv_lora_rank, int64_t qk_rope_dim, double eps)
{
    const c10::DeviceGuard device_guard(x.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(x.dim() == 2 && x.is_contiguous() && x.size(1) == q_lora_rank + kv_lora_rank + qk_rope_dim,
                "fused_split_qk_norm: input must be contiguous [B, q_lora_rank + kv_lora_rank + qk_rope_dim]");
    auto chk = [&](const at::Tensor &t, int64_t n, const char *what) {
        TORCH_CHECK(t.numel() == n && t.is_contiguous() && t.scalar_type() == x.scalar_type(), "fused_split_qk_norm: ", what, " must hold ", n,
                    " values in the input dtype");
    };
    chk(q_weight, q_lora_rank, "q weight"), chk(k_weight, kv_lora_rank, "k weight");
    if (q_bias.has_value()) chk(*q_bias, q_lora_rank, "q bias");
    if (k_bias.has_value()) chk(*k_bias, kv_lora_rank, "k bias");
    const int64_t B = x.size(0);
    at::Tensor q = at::empty({B, q_lora_rank}, x.options()), kn = at::empty({B, kv_lora_rank}, x.options()), kp = at::empty({B, qk_rope_dim}, x.options());
    const int rc = mi_fused_split_qk_norm(x.data_ptr(), B, (int)q_lora_rank, (int)kv_lora_rank, (int)qk_rope_dim, (float)eps, q_weight.data_ptr(),
                                          q_bias.has_value() ? q_bias->data_ptr() : nullptr, k_weight.data_ptr(),
                                          k_bias.has_value() ? k_bias->data_ptr() : nullptr, dtype_code3(x), q.data_ptr(), kn.data_ptr(), kp.data_ptr(),
                                          cur_stream());
    TORCH_CHECK(rc == 0, "mi_fused_split_qk_norm failed with code ", rc);
    return {q, kn, kp};
}

// norm/split_qkv_rmsnorm_rope_pos_cache_half_npu.py:232-407
std::tuple<at::Tensor, at::Tensor, at::Tensor> split_qkv_rmsnorm_rope_pos_cache_half(
    const at::Tensor &input, const at::Tensor &positions, const at::Tensor &cos_sin_cache, int64_t q_hidden_size, int64_t kv_hidden_size,
    int64_t head_dim, std::optional<double> eps, const std::optional<at::Tensor> &q_weight, const std::optional<at::Tensor> &k_weight,
    const std::optional<at::Tensor> &q_bias, const std::optional<at::Tensor> &k_bias, int64_t rope_dim, bool cast_norm_to_bf16)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.is_contiguous() && input.size(1) == q_hidden_size + 2 * kv_hidden_size,
                "split_qkv_rmsnorm_rope_pos_cache_half: input must be contiguous [B, q + 2 kv]");
    const int64_t B = input.size(0);
    TORCH_CHECK(positions.numel() == B && positions.is_contiguous() && (positions.scalar_type() == at::kInt || positions.scalar_type() == at::kLong),
                "positions must be [B], int32 or int64");
    TORCH_CHECK(cos_sin_cache.dim() == 2 && cos_sin_cache.is_contiguous() && cos_sin_cache.size(0) >= 1 && cos_sin_cache.size(1) >= rope_dim,
                "cos_sin_cache must be contiguous [max_seq, rope_dim]");
    TORCH_CHECK((head_dim & (head_dim - 1)) == 0 && head_dim >= 64 && head_dim <= 256 && rope_dim % 16 == 0 && rope_dim <= head_dim,
                "split_qkv_rmsnorm_rope_pos_cache_half: head_dim must be 64, 128 or 256 and rope_dim a multiple of 16 (this build)");
    TORCH_CHECK(q_hidden_size % kv_hidden_size == 0, "q_hidden_size % kv_hidden_size != 0");
    const bool norms = eps.has_value();
    if (norms)
        TORCH_CHECK(q_weight.has_value() && k_weight.has_value() && q_weight->numel() >= head_dim && k_weight->numel() >= head_dim &&
                        q_weight->scalar_type() == input.scalar_type() && k_weight->scalar_type() == input.scalar_type(),
                    "When using RMSNorm (eps is not None), q_weight / k_weight must have at least head_dim elements in the input dtype");
    TORCH_CHECK(q_bias.has_value() == k_bias.has_value(), "q_bias and k_bias go together");
    if (q_bias.has_value())
        TORCH_CHECK(norms && q_bias->numel() >= head_dim && k_bias->numel() >= head_dim && q_bias->scalar_type() == input.scalar_type() &&
                        k_bias->scalar_type() == input.scalar_type(),
                    "bias needs the norm, head_dim elements and the input dtype");
    at::Tensor q = at::empty({B, q_hidden_size}, input.options()), k = at::empty({B, kv_hidden_size}, input.options()),
               v = at::empty({B, kv_hidden_size}, input.options());
    // contiguous copies (no-ops for contiguous arguments) held in locals that outlive the launch
    const at::Tensor qw_c = q_weight.has_value() ? q_weight->contiguous() : at::Tensor(), kw_c = k_weight.has_value() ? k_weight->contiguous() : at::Tensor(),
                     qb_c = q_bias.has_value() ? q_bias->contiguous() : at::Tensor(), kb_c = k_bias.has_value() ? k_bias->contiguous() : at::Tensor();
    auto p = [](const at::Tensor &t) -> const void * { return t.defined() ? t.data_ptr() : nullptr; };
    const int rc = mi_split_qkv_rmsnorm_rope_pos_cache(input.data_ptr(), positions.data_ptr(), positions.scalar_type() == at::kLong, cos_sin_cache.data_ptr(),
                                                       dtype_code3(cos_sin_cache), (int)cos_sin_cache.size(0), cos_sin_cache.stride(0), (int)B,
                                                       (int)q_hidden_size, (int)kv_hidden_size, (int)head_dim, (int)rope_dim, norms, (float)eps.value_or(0.0),
                                                       p(qw_c), p(kw_c), p(qb_c), p(kb_c), cast_norm_to_bf16, dtype_code(input), q.data_ptr(),
                                                       k.data_ptr(), v.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_split_qkv_rmsnorm_rope_pos_cache failed with code ", rc);
    return {q, k, v};
}

// norm/split_qkv_tp_rmsnorm_rope.py:179-288, first launch: (v, qk_var [batch, 2] fp32 = mean(q^2), mean(k^2) of this rank's columns)
std::tuple<at::Tensor, at::Tensor> split_qkv_tp_local_var(const at::Tensor &input, int64_t q_hidden_size, int64_t kv_hidden_size)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.is_contiguous() && input.size(1) == q_hidden_size + 2 * kv_hidden_size,
                "split_qkv_tp_rmsnorm_rope: input must be contiguous [batch, q + 2 kv]");
    const int64_t B = input.size(0);
    at::Tensor v = at::empty({B, kv_hidden_size}, input.options());
    at::Tensor qk_var = at::empty({B, 2}, input.options().dtype(at::kFloat));
    const int rc = mi_split_qkv_tp_var(input.data_ptr(), B, (int)q_hidden_size, (int)kv_hidden_size, dtype_code3(input), v.data_ptr(),
                                       qk_var.data_ptr<float>(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_split_qkv_tp_var failed with code ", rc);
    return {v, qk_var};
}

// second launch, behind the all-reduce of qk_var: (q, k)
std::tuple<at::Tensor, at::Tensor> split_qkv_tp_norm_rope(const at::Tensor &input, const at::Tensor &cos, const at::Tensor &sin, const at::Tensor &qk_var,
                                                          int64_t q_hidden_size, int64_t kv_hidden_size, int64_t head_dim, double eps,
                                                          const at::Tensor &q_weight, const at::Tensor &k_weight, int64_t rotary_dim, double inv_tp_world)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.is_contiguous() && input.size(1) == q_hidden_size + 2 * kv_hidden_size,
                "split_qkv_tp_rmsnorm_rope: input must be contiguous [batch, q + 2 kv]");
    const int64_t B = input.size(0);
    TORCH_CHECK((head_dim & (head_dim - 1)) == 0, "head_dim must be a power of two");              // reference :229-230
    TORCH_CHECK(q_hidden_size % kv_hidden_size == 0, "q_hidden_size % kv_hidden_size != 0");         // reference :231
    TORCH_CHECK(cos.numel() == B * rotary_dim && sin.numel() == B * rotary_dim && cos.is_contiguous() && sin.is_contiguous() &&
                    cos.scalar_type() == input.scalar_type() && sin.scalar_type() == input.scalar_type(),
                "split_qkv_tp_rmsnorm_rope: cos / sin must be contiguous [batch, rotary_dim] in the input dtype");
    TORCH_CHECK(q_weight.numel() == q_hidden_size && k_weight.numel() == kv_hidden_size && q_weight.is_contiguous() && k_weight.is_contiguous() &&
                    q_weight.scalar_type() == input.scalar_type() && k_weight.scalar_type() == input.scalar_type(),
                "split_qkv_tp_rmsnorm_rope: weights must be [q_hidden_size] / [kv_hidden_size] in the input dtype");
    TORCH_CHECK(qk_var.numel() == 2 * B && qk_var.is_contiguous() && qk_var.scalar_type() == at::kFloat, "qk_var must be float32 [batch, 2]");
    at::Tensor q = at::empty({B, q_hidden_size}, input.options()), k = at::empty({B, kv_hidden_size}, input.options());
    const int rc = mi_split_qkv_tp_norm_rope(input.data_ptr(), cos.data_ptr(), sin.data_ptr(), qk_var.data_ptr<float>(), B, (int)q_hidden_size,
                                             (int)kv_hidden_size, (int)head_dim, (int)rotary_dim, (float)eps, (float)inv_tp_world, q_weight.data_ptr(),
                                             k_weight.data_ptr(), dtype_code3(input), q.data_ptr(), k.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_split_qkv_tp_norm_rope failed with code ", rc);
    return {q, k};
}

// split [q | gate] + K + V, Gemma RMSNorm + neox RoPE; arguments as split_qkvgate_gemma_rmsnorm_rope (norm/split_qkv_rmsnorm_rope.py:686-745)
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> split_qkvgate_gemma_rmsnorm_rope(
    const at::Tensor &input, const at::Tensor &sin, const at::Tensor &cos, int64_t q_hidden_size, int64_t kv_hidden_size,
    int64_t head_dim, int64_t rope_dim, double eps, const at::Tensor &q_weight, const at::Tensor &k_weight)
{
    const c10::DeviceGuard device_guard(input.device());   // launches and the current stream follow the tensor's GPU
    TORCH_CHECK(input.dim() == 2 && input.is_contiguous(), "split_qkvgate_gemma_rmsnorm_rope: input must be contiguous [batch, 2q+2kv]");
    TORCH_CHECK((head_dim & (head_dim - 1)) == 0, "head_dim must be a power of two");        // reference :700-701
    TORCH_CHECK(q_hidden_size % kv_hidden_size == 0, "q_hidden_size % kv_hidden_size != 0");   // reference :702
    TORCH_CHECK(input.size(1) == 2 * q_hidden_size + 2 * kv_hidden_size, "split_qkvgate_gemma_rmsnorm_rope: input width");
    const int64_t B = input.size(0);
    TORCH_CHECK(sin.numel() == B * rope_dim && cos.numel() == B * rope_dim && sin.is_contiguous() && cos.is_contiguous() &&
                    sin.scalar_type() == input.scalar_type() && cos.scalar_type() == input.scalar_type(),
                "split_qkvgate_gemma_rmsnorm_rope: sin/cos must be contiguous [batch, ..., rope_dim] in the input dtype");
    TORCH_CHECK(q_weight.numel() == head_dim && k_weight.numel() == head_dim && q_weight.is_contiguous() && k_weight.is_contiguous() &&
                    q_weight.scalar_type() == input.scalar_type() && k_weight.scalar_type() == input.scalar_type(),
                "split_qkvgate_gemma_rmsnorm_rope: norm weights must be [head_dim] in the input dtype");
    at::Tensor q = at::empty({B, q_hidden_size}, input.options()), k = at::empty({B, kv_hidden_size}, input.options()),
               v = at::empty({B, kv_hidden_size}, input.options()), gate = at::empty({B, q_hidden_size}, input.options());
    const int rc = mi_split_qkvgate_gemma_rmsnorm_rope(input.data_ptr(), sin.data_ptr(), cos.data_ptr(), (int)B, (int)q_hidden_size,
                                                       (int)kv_hidden_size, (int)head_dim, (int)rope_dim, (float)eps, q_weight.data_ptr(),
                                                       k_weight.data_ptr(), dtype_code(input), q.data_ptr(), k.data_ptr(), v.data_ptr(),
                                                       gate.data_ptr(), cur_stream());
    TORCH_CHECK(rc == 0, "mi_split_qkvgate_gemma_rmsnorm_rope failed with code ", rc);
    return {q, k, v, gate};
}

// torch.ops.npu.mla_preprocess: same schema as the reference (csrc/pytorch_extensions.cpp:105-115; host
// csrc/mla_preprocess/op_host/mla_preprocess.cpp:623-704).  MI355X layouts: wdqkv int8 [2112, hidden] and wuq int8
// [q_heads*192, 1536] are plain row-major (output channel major, K contiguous) instead of the Ascend NZ fractal format;
// wuk [q_heads, 128, 512]; kv_cache [blocks, block_size, 1, 512] + kv_cache_rope [blocks, block_size, 1, 64]
// (cache_mode "krope_ctkv": [slot][dim]; "nzcache": per block [dim/16][slot in block][16]; "int8_nzcache": k_nope quantised with
// ctkv_scale into [dim/32][slot in block][32] and q_out0 as int8 quantised with q_nope_scale[head] -- the cache block is the
// unit of the NZ layouts, weights stay row-major).  gamma0 / beta0 are accepted and unused, as in the reference's bf16 kernel (stage 1 is
// quant-only).  Every stage is a HIP kernel of csrc/kernels/mla_gemm.hip (the two INT8 GEMMs and the per-head BMM, hand-written
// skinny split-K / weight-streaming kernels) or csrc/kernels/mla_preprocess.hip (quant, dequant + split + RMSNorm + RoPE + cache).
// wuk is consumed K-contiguous: its [q_heads, 512, 128] transpose is made once per weight tensor and kept (the reference casts
// its weights to the NZ format once as well).
static at::Tensor prepared_wuk(const at::Tensor &wuk, at::ScalarType dtype)
{
    // Keyed on the CALLER's tensor (address, version, target dtype): a dtype conversion happens inside the make step, so a wuk
    // stored in another dtype than the activations still hits.  Inference tensors carry no version counter (_version() throws):
    // they are recorded as version -1 and matched on (address, held storage, shape, dtype) alone.
    struct Entry {
        at::Tensor src, t;      // the source is held: its storage cannot be recycled for another weight while the entry lives
        int64_t version;
    };
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, Entry> cache;
    const int64_t ver = wuk.is_inference() ? -1 : (int64_t)wuk._version();
    const auto key = std::make_pair((const void *)wuk.data_ptr(), (int)dtype);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end() && it->second.version == ver && it->second.src.sizes() == wuk.sizes() &&
        it->second.src.scalar_type() == wuk.scalar_type())
        return it->second.t;
    if (cache.size() >= 256) cache.clear();
    at::Tensor t = wuk.to(dtype).transpose(1, 2).contiguous();
    cache[key] = Entry{wuk, t, ver};
    return t;
}

std::tuple<at::Tensor &, at::Tensor &, at::Tensor &, at::Tensor &> mla_preprocess(
    const at::Tensor &hiddenState, const at::Tensor &gamma0, const at::Tensor &beta0, const at::Tensor &wdqkv,
    const at::Tensor &descale0, const at::Tensor &gamma1, const at::Tensor &beta1, const at::Tensor &wuq,
    const at::Tensor &descale1, const at::Tensor &gamma2, const at::Tensor &cos, const at::Tensor &sin, const at::Tensor &wuk,
    const at::Tensor &kv_cache, const at::Tensor &kv_cache_rope, const at::Tensor &slotmapping, const at::Tensor &quant_scale0,
    const at::Tensor &quant_offset0, const at::Tensor &bias0, const at::Tensor &quant_scale1, const at::Tensor &quant_offset1,
    const at::Tensor &bias1, const std::optional<at::Tensor> &ctkv_scale, const std::optional<at::Tensor> &q_nope_scale,
    std::optional<c10::string_view> cache_mode, std::optional<c10::string_view> quant_mode, at::Tensor &q_out0,
    at::Tensor &kv_cache_out0, at::Tensor &q_out1, at::Tensor &kv_cache_out1)
{
    const c10::DeviceGuard device_guard(hiddenState.device());   // launches and the current stream follow the tensor's GPU
    (void)gamma0, (void)beta0;
    // cache_mode (csrc/mla_preprocess/op_host/mla_preprocess.cpp:605-606,634): krope_ctkv = 1 (default), int8_nzcache = 2, nzcache = 3
    const c10::string_view cmode = cache_mode.value_or("krope_ctkv");
    TORCH_CHECK(cmode == "krope_ctkv" || cmode == "int8_nzcache" || cmode == "nzcache", "Unsupported cache_mode value: '", cmode, "'");
    const int cmode_i = cmode == "krope_ctkv" ? 1 : cmode == "int8_nzcache" ? 2 : 3;
    // quant_mode: "per_tensor_quant_asymm" (the mode the reference tests cover) or "per_token_quant_symm" -- the reference's DEFAULT
    // when the argument is omitted (csrc/mla_preprocess/op_host/mla_preprocess.cpp:634-635), so it is the default here too
    const c10::string_view qmode = quant_mode.value_or("per_token_quant_symm");
    TORCH_CHECK(qmode == "per_tensor_quant_asymm" || qmode == "per_token_quant_symm", "Unsupported quant_mode value: '", qmode, "'");
    const bool per_token = qmode == "per_token_quant_symm";
    TORCH_CHECK(hiddenState.dim() == 2 && hiddenState.is_contiguous(), "hiddenState must be contiguous [tokens, hidden]");
    const int64_t N = hiddenState.size(0), hidden = hiddenState.size(1);
    TORCH_CHECK(N <= 1024, "mla_preprocess: tokenNum <= 1024 (csrc/mla_preprocess/README.md)");
    TORCH_CHECK(wdqkv.scalar_type() == at::kChar && wdqkv.dim() == 2 && wdqkv.size(0) == 2112 && wdqkv.size(1) == hidden &&
                    wdqkv.is_contiguous(), "wdqkv must be int8 [2112, hidden] row-major");
    TORCH_CHECK(wuq.scalar_type() == at::kChar && wuq.dim() == 2 && wuq.size(1) == 1536 && wuq.size(0) % 192 == 0 && wuq.is_contiguous(),
                "wuq must be int8 [q_heads*192, 1536] row-major");
    const int64_t Hq = wuq.size(0) / 192;
    TORCH_CHECK(wuk.dim() == 3 && wuk.size(0) == Hq && wuk.size(1) == 128 && wuk.size(2) == 512, "wuk must be [q_heads, 128, 512]");
    TORCH_CHECK(descale0.scalar_type() == at::kFloat && descale0.numel() == 2112 && descale1.scalar_type() == at::kFloat &&
                    descale1.numel() == Hq * 192, "descale0 / descale1 must be float32 [2112] / [q_heads*192]");
    TORCH_CHECK(slotmapping.scalar_type() == at::kInt && slotmapping.numel() == N, "slotmapping must be int32 [tokens]");
    TORCH_CHECK(cos.numel() == N * 64 && sin.numel() == N * 64 && cos.is_contiguous() && sin.is_contiguous(), "cos / sin must be [tokens, 64]");
    TORCH_CHECK(kv_cache.is_contiguous() && kv_cache.size(-1) == 512 && kv_cache_rope.is_contiguous() && kv_cache_rope.size(-1) == 64,
                "kv_cache [..., 512] and kv_cache_rope [..., 64] must be contiguous");
    TORCH_CHECK(q_out0.is_contiguous() && q_out0.numel() == N * Hq * 512 && q_out1.is_contiguous() && q_out1.numel() == N * Hq * 64,
                "q_out0 [tokens, q_heads, 512] / q_out1 [tokens, q_heads, 64]");
    int64_t block_size = 0;
    if (cmode_i != 1) {        // the NZ layouts are defined per cache block: [blocks, block_size, 1, dim]
        TORCH_CHECK(kv_cache.dim() >= 2 && kv_cache_rope.dim() == kv_cache.dim() && kv_cache_rope.size(1) == kv_cache.size(1),
                    "nzcache modes need kv_cache [blocks, block_size, ..., 512] and kv_cache_rope with the same block_size");
        block_size = kv_cache.size(1);
    }
    if (cmode_i == 2) {
        TORCH_CHECK(kv_cache.scalar_type() == at::kChar && q_out0.scalar_type() == at::kChar,
                    "cache_mode='int8_nzcache': kv_cache and q_out0 must be int8");
        TORCH_CHECK(ctkv_scale.has_value() && ctkv_scale->numel() == 1 && ctkv_scale->scalar_type() == hiddenState.scalar_type() &&
                        q_nope_scale.has_value() && q_nope_scale->numel() == Hq && q_nope_scale->is_contiguous() &&
                        q_nope_scale->scalar_type() == hiddenState.scalar_type(),
                    "cache_mode='int8_nzcache' needs ctkv_scale [1] and q_nope_scale [q_heads] in the input dtype");
    } else {
        TORCH_CHECK(kv_cache.scalar_type() == hiddenState.scalar_type() && q_out0.scalar_type() == hiddenState.scalar_type(),
                    "kv_cache / q_out0 must have the input dtype");
    }
    TORCH_CHECK(kv_cache_rope.scalar_type() == hiddenState.scalar_type() && q_out1.scalar_type() == hiddenState.scalar_type(),
                "kv_cache_rope / q_out1 must have the input dtype");
    const int dt = dtype_code(hiddenState);
    auto dev = hiddenState.device();
    void *st = cur_stream();
    TORCH_CHECK(per_token ||
                    (quant_scale0.numel() == 1 && quant_scale1.numel() == 1 && quant_scale0.scalar_type() == hiddenState.scalar_type() &&
                     quant_scale1.scalar_type() == hiddenState.scalar_type() && quant_offset0.numel() == 1 && quant_offset1.numel() == 1 &&
                     quant_offset0.scalar_type() == at::kChar && quant_offset1.scalar_type() == at::kChar),
                "quant_scale0/1 must be [1] in the input dtype, quant_offset0/1 int8 [1]");
    // five launches on the caller's stream, no library GEMM (stage order of mla_preprocess_mix_bf16.hpp):
    //   quant -> INT8 GEMM1 split-K (partial products) -> sum + dequant / RMSNorm / RoPE / cache write / requant
    //   -> INT8 GEMM2 with the dequant epilogue -> per-head BMM + RoPE of the positional columns
    auto i8 = at::dtype(at::kChar).device(dev);
    at::Tensor a8 = at::empty({N, hidden}, i8);
    const int parts = mi_mla_pre_gemm_i8_partials((int)hidden);                 // split-K partial products, summed by pre_mid
    at::Tensor c1 = at::empty({parts, N, 2112}, at::dtype(at::kInt).device(dev));
    // per-token mode: every row is quantised against its own maximum and carries its scale to the dequant of the following GEMM
    at::Tensor tok0, tok1;
    if (per_token) tok0 = at::empty({N}, at::dtype(at::kFloat).device(dev)), tok1 = at::empty({N}, at::dtype(at::kFloat).device(dev));
    at::Tensor q8 = at::empty({N, 1536}, i8);
    auto iptr = [](const at::Tensor &t) -> const int32_t * { return t.numel() ? t.data_ptr<int32_t>() : nullptr; };
    // Opt-in (MI_MLA_PRE_ONE_LAUNCH=1), decode sizes: the whole op as ONE launch (stage bodies of the four launches behind grid barriers
    // inside one grid; bit-identical; measured no faster than the four launches, csrc/kernels/mla_gemm.hip).  The barrier words come
    // from a per-device ring: a buffer serves one call in flight, calls on different streams get different buffers.
    const char *ol_env = getenv("MI_MLA_PRE_ONE_LAUNCH");      // read per call: the tests compare both forms in one process
    const bool one_launch = ol_env && atoi(ol_env) != 0;
    if (one_launch && N > 0) {
        constexpr int kRing = 16;
        static std::mutex mu;
        // (never destroyed: a static tensor freed at process exit would outlive the HIP context)
        static auto &rings = *new std::map<int, std::pair<at::Tensor, uint64_t>>();
        const int64_t words = (int64_t)mi_mla_preprocess_one_launch_sync_words();
        uint32_t *sync = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto &r = rings[dev.index()];
            if (!r.first.defined()) r.first = at::zeros({kRing, words}, at::dtype(at::kInt).device(dev));
            sync = (uint32_t *)r.first.data_ptr<int32_t>() + (r.second++ % kRing) * words;
        }
        at::Tensor wuk_t1 = prepared_wuk(wuk, hiddenState.scalar_type());
        const int rc = mi_mla_preprocess_one_launch(
            hiddenState.data_ptr(), (int)N, (int)hidden, per_token ? nullptr : quant_scale0.data_ptr(),
            per_token ? nullptr : (const int8_t *)quant_offset0.data_ptr(), (int8_t *)a8.data_ptr(), per_token ? tok0.data_ptr<float>() : nullptr,
            (const int8_t *)wdqkv.data_ptr(), c1.data_ptr<int32_t>(), iptr(bias0), descale0.data_ptr<float>(), gamma1.data_ptr(), beta1.data_ptr(),
            gamma2.data_ptr(), cos.data_ptr(), sin.data_ptr(), slotmapping.data_ptr<int32_t>(), per_token ? nullptr : quant_scale1.data_ptr(),
            per_token ? nullptr : (const int8_t *)quant_offset1.data_ptr(), 1e-6f, (int8_t *)q8.data_ptr(), kv_cache.data_ptr(),
            kv_cache_rope.data_ptr(), per_token ? tok1.data_ptr<float>() : nullptr, cmode_i, (int)block_size,
            cmode_i == 2 ? ctkv_scale->data_ptr() : nullptr, (const int8_t *)wuq.data_ptr(), (int)Hq, per_token ? nullptr : iptr(bias1),
            descale1.data_ptr<float>(), wuk_t1.data_ptr(), q_out0.data_ptr(), q_out1.data_ptr(), cmode_i == 2 ? q_nope_scale->data_ptr() : nullptr,
            per_token ? 1 : 0, dt, sync, st);
        TORCH_CHECK(rc == 0 || rc == MI_SGL_ENOTAPPLICABLE, "mi_mla_preprocess_one_launch failed with code ", rc);
        if (rc == 0) {
            if (kv_cache_out0.data_ptr() != kv_cache.data_ptr()) kv_cache_out0.copy_(kv_cache);
            if (kv_cache_out1.data_ptr() != kv_cache_rope.data_ptr()) kv_cache_out1.copy_(kv_cache_rope);
            return {q_out0, kv_cache_out0, q_out1, kv_cache_out1};
        }
    }
    if (per_token) {
        TORCH_CHECK(0 == mi_mla_pre_quant_token(hiddenState.data_ptr(), (int)N, (int)hidden, dt, (int8_t *)a8.data_ptr(), tok0.data_ptr<float>(), st),
                    "mi_mla_pre_quant_token failed");
    } else {
        TORCH_CHECK(0 == mi_mla_pre_quant(hiddenState.data_ptr(), quant_scale0.data_ptr(), (const int8_t *)quant_offset0.data_ptr(), N * hidden, dt,
                                          (int8_t *)a8.data_ptr(), st), "mi_mla_pre_quant failed");
    }
    TORCH_CHECK(0 == mi_mla_pre_gemm_i8((const int8_t *)a8.data_ptr(), (int)N, (int)hidden, (const int8_t *)wdqkv.data_ptr(), 2112, 0,
                                        c1.data_ptr<int32_t>(), nullptr, nullptr, nullptr, nullptr, dt, st), "mi_mla_pre_gemm_i8 (GEMM1) failed");
    TORCH_CHECK(0 == mi_mla_pre_mid(c1.data_ptr<int32_t>(), parts, iptr(bias0), descale0.data_ptr<float>(), gamma1.data_ptr(), beta1.data_ptr(),
                                    gamma2.data_ptr(), cos.data_ptr(), sin.data_ptr(), slotmapping.data_ptr<int32_t>(),
                                    per_token ? nullptr : quant_scale1.data_ptr(), per_token ? nullptr : (const int8_t *)quant_offset1.data_ptr(),
                                    1e-6f, (int)N, dt, (int8_t *)q8.data_ptr(), kv_cache.data_ptr(), kv_cache_rope.data_ptr(),
                                    per_token ? tok0.data_ptr<float>() : nullptr, per_token ? tok1.data_ptr<float>() : nullptr, cmode_i,
                                    (int)block_size, cmode_i == 2 ? ctkv_scale->data_ptr() : nullptr, st),
                "mi_mla_pre_mid failed");
    at::Tensor wuk_t = prepared_wuk(wuk, hiddenState.scalar_type());
    const void *qns = cmode_i == 2 ? q_nope_scale->data_ptr() : nullptr;
    // GEMM2 + per-head BMM + RoPE: one launch, a workgroup per head, the GEMM2 output stays in LDS (MI_MLA_PRE_FUSED=0: the two-launch
    // form with the GEMM2 output materialised in global memory; bit-identical)
    static const bool fused = !(getenv("MI_MLA_PRE_FUSED") && atoi(getenv("MI_MLA_PRE_FUSED")) == 0);
    if (fused) {
        TORCH_CHECK(0 == mi_mla_pre_gemm2_bmm_rope((const int8_t *)q8.data_ptr(), (int)N, (const int8_t *)wuq.data_ptr(), (int)Hq,
                                                   per_token ? nullptr : iptr(bias1), descale1.data_ptr<float>(),
                                                   per_token ? tok1.data_ptr<float>() : nullptr, wuk_t.data_ptr(), cos.data_ptr(), sin.data_ptr(),
                                                   dt, q_out0.data_ptr(), q_out1.data_ptr(), qns, st), "mi_mla_pre_gemm2_bmm_rope failed");
    } else {
        at::Tensor y2 = at::empty({N, Hq * 192}, hiddenState.options());          // GEMM2 output in the I/O dtype (golden :95-107)
        TORCH_CHECK(0 == mi_mla_pre_gemm_i8((const int8_t *)q8.data_ptr(), (int)N, 1536, (const int8_t *)wuq.data_ptr(), (int)(Hq * 192), 1,
                                            nullptr, per_token ? nullptr : iptr(bias1), descale1.data_ptr<float>(),
                                            per_token ? tok1.data_ptr<float>() : nullptr, y2.data_ptr(), dt, st), "mi_mla_pre_gemm_i8 (GEMM2) failed");
        TORCH_CHECK(0 == mi_mla_pre_bmm_rope(y2.data_ptr(), (int)N, (int)Hq, wuk_t.data_ptr(), cos.data_ptr(), sin.data_ptr(), dt,
                                             q_out0.data_ptr(), q_out1.data_ptr(), qns, st), "mi_mla_pre_bmm_rope failed");
    }
    if (kv_cache_out0.data_ptr() != kv_cache.data_ptr()) kv_cache_out0.copy_(kv_cache);
    if (kv_cache_out1.data_ptr() != kv_cache_rope.data_ptr()) kv_cache_out1.copy_(kv_cache_rope);
    return {q_out0, kv_cache_out0, q_out1, kv_cache_out1};
}

}  // namespace npu_kernel
}  // namespace sglang

TORCH_LIBRARY_FRAGMENT(npu, m)
{
    m.def("sgl_kernel_npu_version() -> str", &sglang::npu_kernel::sgl_kernel_npu_version);
    m.def("decode_mla(Tensor q, Tensor k_nope_buffer, Tensor k_rope_buffer, Tensor(a!) att_out, Tensor kv_seq_lens, "
          "float sm_scale, int page_size, Tensor block_table, int num_splits=0) -> ()");
    m.def("decode_mla_plan(Tensor kv_seq_lens, int num_kv_heads=1) -> Tensor");
    m.def("clear_mla_plan_cache", &sglang::npu_kernel::clear_mla_plan_cache);      // no tensor arguments: catch-all kernel
    m.def("decode_mla_planned(Tensor q, Tensor k_nope_buffer, Tensor k_rope_buffer, Tensor(a!) att_out, Tensor kv_seq_lens, "
          "float sm_scale, int page_size, Tensor block_table, Tensor plan) -> ()");
    m.def("decode_gqa(Tensor q, Tensor k_buffer, Tensor v_buffer, Tensor(a!) att_out, Tensor kv_seq_lens, "
          "float sm_scale, int page_size, Tensor block_table, int num_splits=0) -> ()");
    m.def("mla_preprocess(Tensor hiddenState, Tensor gamma0, Tensor beta0, Tensor wdqkv, "
          "Tensor descale0, Tensor gamma1, Tensor beta1, Tensor wuq, "
          "Tensor descale1, Tensor gamma2, Tensor cos, Tensor sin, Tensor wuk,"
          "Tensor kv_cache, Tensor kv_cache_rope, Tensor slotmapping, "
          "Tensor quant_scale0, Tensor quant_offset0, Tensor bias0, "
          "Tensor quant_scale1, Tensor quant_offset1, Tensor bias1, *, "
          "Tensor? ctkv_scale=None, Tensor? q_nope_scale=None, "
          "str? cache_mode=None, str? quant_mode=None, "
          "Tensor(a!) q_out0, Tensor(b!) kv_cache_out0, Tensor(c!) q_out1, Tensor(d!) kv_cache_out1) "
          "-> (Tensor(a!), Tensor(b!), Tensor(c!), Tensor(d!))");
    m.def("swiglu_quant(Tensor x, Tensor group_list, int group_list_type, bool need_quant=True, bool do_limit=False, "
          "float limit=7.0) -> (Tensor, Tensor)");
    m.def("add_rmsnorm_bias(Tensor input, Tensor? residual, Tensor norm_weight, Tensor? norm_bias, float eps, "
          "Tensor? quant_scale=None, Tensor? quant_offset=None, bool gemma=False) -> (Tensor, Tensor)");
    m.def("fused_rope_qk_mqa(Tensor query, Tensor key, Tensor cos_sin, int rotary_dim, bool is_neox_style) -> (Tensor, Tensor)");
    m.def("split_qkv_rmsnorm_rope(Tensor input, Tensor sin, Tensor cos, int q_hidden_size, int kv_hidden_size, int head_dim, "
          "float? eps=None, Tensor? q_weight=None, Tensor? k_weight=None, Tensor? q_bias=None, Tensor? k_bias=None, "
          "bool is_neox_style=True) -> (Tensor, Tensor, Tensor)");
    m.def("l1_norm(Tensor input) -> Tensor");
    m.def("rmsnorm_without_weight(Tensor x, float eps) -> Tensor");
    m.def("fused_variance(Tensor x) -> Tensor");
    m.def("fused_rsqrt_mul(Tensor x, Tensor variance, Tensor weight, float eps=1e-6) -> Tensor");
    m.def("fused_scale_shift(Tensor x, Tensor scale, Tensor shift, float scale_constant=1.0) -> Tensor");
    m.def("split_qkv_rmsnorm_mrope(Tensor qkv, Tensor q_weight, Tensor k_weight, Tensor cos_sin, int num_q_heads, int num_kv_heads, int head_size, "
          "float eps, int[] mrope_section, bool is_interleaved, int? rope_dim=None, Tensor? q_bias=None, Tensor? k_bias=None, "
          "bool has_gate=False) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("split_qkv_rmsnorm_rope_pos_cache_half(Tensor input, Tensor positions, Tensor cos_sin_cache, int q_hidden_size, int kv_hidden_size, "
          "int head_dim, float? eps, Tensor? q_weight, Tensor? k_weight, Tensor? q_bias, Tensor? k_bias, int rope_dim, "
          "bool cast_norm_to_bf16) -> (Tensor, Tensor, Tensor)");
    m.def("attention_sinks(Tensor query, Tensor k_cache, Tensor v_cache, Tensor sinks, Tensor block_tables, Tensor kv_lens, float scale, "
          "int sliding_window_size, int q_head_num, int k_head_num, Tensor? bt_rows=None) -> Tensor");
    m.def("swiglu_oai_quant(Tensor x, float alpha, float limit, bool need_quant=True, Tensor? group_list=None, int? group_list_type=None) -> (Tensor, Tensor)");
    m.def("situ_and_mul(Tensor x, Tensor? group_list, int? group_list_type, float beta, float? linear_beta, bool need_quant) -> (Tensor, Tensor)");
    m.def("fia_blockq_sparse_prefill(Tensor q, Tensor k_cache, Tensor v_cache, Tensor topk_idx, Tensor seq_lens, Tensor per_query_req, "
          "Tensor req_to_token, int block_size, float sm_scale, Tensor? block_table_out=None, Tensor? actual_kvlen_out=None) -> Tensor");
    m.def("attn_residual_mix(Tensor prefix_sum, Tensor bank, int num_valid_blocks, Tensor combined_weight, float variance_epsilon) -> Tensor");
    m.def("mul_add(Tensor routed_input, Tensor shared_input, float scaling_factor) -> Tensor");
    m.def("zero_experts_compute_identity(Tensor(a!) expert_indices, Tensor(b!) expert_scales, int num_experts, Tensor hidden_states, "
          "int identity_mask_value=0) -> Tensor");
    m.def("swiglu_oai(Tensor hidden_states, int dim, float gemm1_alpha, float gemm1_clamp_limit) -> Tensor");
    m.def("fused_split_qk_norm(Tensor x, Tensor q_weight, Tensor? q_bias, Tensor k_weight, Tensor? k_bias, int q_lora_rank, int kv_lora_rank, "
          "int qk_rope_dim, float eps=1e-6) -> (Tensor, Tensor, Tensor)");
    m.def("split_qkv_tp_local_var(Tensor input, int q_hidden_size, int kv_hidden_size) -> (Tensor, Tensor)");
    m.def("split_qkv_tp_norm_rope(Tensor input, Tensor cos, Tensor sin, Tensor qk_var, int q_hidden_size, int kv_hidden_size, int head_dim, "
          "float eps, Tensor q_weight, Tensor k_weight, int rotary_dim, float inv_tp_world) -> (Tensor, Tensor)");
    m.def("split_qkvgate_gemma_rmsnorm_rope(Tensor input, Tensor sin, Tensor cos, int q_hidden_size, int kv_hidden_size, int head_dim, "
          "int rope_dim, float eps, Tensor q_weight, Tensor k_weight) -> (Tensor, Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(npu, CUDA, m)
{
    m.impl("decode_mla", TORCH_FN(sglang::npu_kernel::decode_mla));
    m.impl("decode_mla_plan", TORCH_FN(sglang::npu_kernel::decode_mla_plan));
    m.impl("decode_mla_planned", TORCH_FN(sglang::npu_kernel::decode_mla_planned));
    m.impl("decode_gqa", TORCH_FN(sglang::npu_kernel::decode_gqa));
    m.impl("mla_preprocess", TORCH_FN(sglang::npu_kernel::mla_preprocess));
    m.impl("swiglu_quant", TORCH_FN(sglang::npu_kernel::swiglu_quant));
    m.impl("add_rmsnorm_bias", TORCH_FN(sglang::npu_kernel::add_rmsnorm_bias));
    m.impl("fused_rope_qk_mqa", TORCH_FN(sglang::npu_kernel::fused_rope_qk_mqa));
    m.impl("split_qkv_rmsnorm_rope", TORCH_FN(sglang::npu_kernel::split_qkv_rmsnorm_rope));
    m.impl("split_qkvgate_gemma_rmsnorm_rope", TORCH_FN(sglang::npu_kernel::split_qkvgate_gemma_rmsnorm_rope));
    m.impl("l1_norm", TORCH_FN(sglang::npu_kernel::l1_norm));
    m.impl("rmsnorm_without_weight", TORCH_FN(sglang::npu_kernel::rmsnorm_without_weight));
    m.impl("fused_variance", TORCH_FN(sglang::npu_kernel::fused_variance));
    m.impl("fused_rsqrt_mul", TORCH_FN(sglang::npu_kernel::fused_rsqrt_mul));
    m.impl("fused_scale_shift", TORCH_FN(sglang::npu_kernel::fused_scale_shift));
    m.impl("split_qkv_rmsnorm_mrope", TORCH_FN(sglang::npu_kernel::split_qkv_rmsnorm_mrope));
    m.impl("split_qkv_rmsnorm_rope_pos_cache_half", TORCH_FN(sglang::npu_kernel::split_qkv_rmsnorm_rope_pos_cache_half));
    m.impl("attention_sinks", TORCH_FN(sglang::npu_kernel::attention_sinks));
    m.impl("swiglu_oai_quant", TORCH_FN(sglang::npu_kernel::swiglu_oai_quant));
    m.impl("situ_and_mul", TORCH_FN(sglang::npu_kernel::situ_and_mul));
    m.impl("fia_blockq_sparse_prefill", TORCH_FN(sglang::npu_kernel::fia_blockq_sparse_prefill));
    m.impl("attn_residual_mix", TORCH_FN(sglang::npu_kernel::attn_residual_mix));
    m.impl("mul_add", TORCH_FN(sglang::npu_kernel::mul_add));
    m.impl("zero_experts_compute_identity", TORCH_FN(sglang::npu_kernel::zero_experts_compute_identity));
    m.impl("swiglu_oai", TORCH_FN(sglang::npu_kernel::swiglu_oai));
    m.impl("fused_split_qk_norm", TORCH_FN(sglang::npu_kernel::fused_split_qk_norm));
    m.impl("split_qkv_tp_local_var", TORCH_FN(sglang::npu_kernel::split_qkv_tp_local_var));
    m.impl("split_qkv_tp_norm_rope", TORCH_FN(sglang::npu_kernel::split_qkv_tp_norm_rope));
}
