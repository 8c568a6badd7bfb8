// torch.ops.npu.* registry for MI355X.
// Mirrors the reference registration unit csrc/pytorch_extensions.cpp (schemas :24-201 under TORCH_LIBRARY_FRAGMENT(npu, m),
// implementations :205-316) for the hot-path operators: the namespace stays `npu` so SGLang's call sites
// (`torch.ops.npu.<op>`) are unchanged; the dispatch key is CUDA (= HIP on ROCm) instead of PrivateUse1.
// The implementations are thin host functions (namespace sglang::npu_kernel, like include/sgl_kenel_npu_ops.h:14-239)
// that validate arguments, allocate outputs and call the C-ABI of include/mi_sgl_kernels.h on the current stream.
#include <ATen/ATen.h>
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

std::string sgl_kernel_npu_version() { return std::string("sgl-kernel-npu_amd 0.1 (") + mi_sgl_kernels_version() + ")"; }

// Paged MLA decode, same argument meaning as the reference Python entry point decode_mla
// (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:166-175); writes att_out in place.
void decode_mla(const at::Tensor &q, const at::Tensor &k_nope_buffer, const at::Tensor &k_rope_buffer, at::Tensor &att_out,
                const at::Tensor &kv_seq_lens, double sm_scale, int64_t page_size, const at::Tensor &block_table,
                int64_t num_splits)
{
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
    const int max_len = (int)std::min<int64_t>(block_table.size(1) * page_size, INT32_MAX);   // upper bound, no host sync
    int splits = (int)num_splits;
    if (splits <= 0) splits = mi_mla_decode_num_splits(B, Hq, Hkv, max_len);
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

// SwiGLU + per-row INT8 quantisation; same arguments / returns as swiglu_quant (activation/swiglu_quant.py:87-127).
std::tuple<at::Tensor, at::Tensor> swiglu_quant(const at::Tensor &x, const at::Tensor &group_list, int64_t group_list_type,
                                                bool need_quant, bool do_limit, double limit)
{
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

// split QKV + per-head RMSNorm + RoPE; arguments as split_qkv_rmsnorm_rope (norm/split_qkv_rmsnorm_rope.py:374-438).
std::tuple<at::Tensor, at::Tensor, at::Tensor> split_qkv_rmsnorm_rope(
    const at::Tensor &input, const at::Tensor &sin, const at::Tensor &cos, int64_t q_hidden_size, int64_t kv_hidden_size,
    int64_t head_dim, std::optional<double> eps, const std::optional<at::Tensor> &q_weight,
    const std::optional<at::Tensor> &k_weight, const std::optional<at::Tensor> &q_bias, const std::optional<at::Tensor> &k_bias,
    bool is_neox_style)
{
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

}  // namespace npu_kernel
}  // namespace sglang

TORCH_LIBRARY_FRAGMENT(npu, m)
{
    m.def("sgl_kernel_npu_version() -> str", &sglang::npu_kernel::sgl_kernel_npu_version);
    m.def("decode_mla(Tensor q, Tensor k_nope_buffer, Tensor k_rope_buffer, Tensor(a!) att_out, Tensor kv_seq_lens, "
          "float sm_scale, int page_size, Tensor block_table, int num_splits=0) -> ()");
    m.def("swiglu_quant(Tensor x, Tensor group_list, int group_list_type, bool need_quant=True, bool do_limit=False, "
          "float limit=7.0) -> (Tensor, Tensor)");
    m.def("add_rmsnorm_bias(Tensor input, Tensor? residual, Tensor norm_weight, Tensor? norm_bias, float eps, "
          "Tensor? quant_scale=None, Tensor? quant_offset=None, bool gemma=False) -> (Tensor, Tensor)");
    m.def("split_qkv_rmsnorm_rope(Tensor input, Tensor sin, Tensor cos, int q_hidden_size, int kv_hidden_size, int head_dim, "
          "float? eps=None, Tensor? q_weight=None, Tensor? k_weight=None, Tensor? q_bias=None, Tensor? k_bias=None, "
          "bool is_neox_style=True) -> (Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(npu, CUDA, m)
{
    m.impl("decode_mla", TORCH_FN(sglang::npu_kernel::decode_mla));
    m.impl("swiglu_quant", TORCH_FN(sglang::npu_kernel::swiglu_quant));
    m.impl("add_rmsnorm_bias", TORCH_FN(sglang::npu_kernel::add_rmsnorm_bias));
    m.impl("split_qkv_rmsnorm_rope", TORCH_FN(sglang::npu_kernel::split_qkv_rmsnorm_rope));
}
