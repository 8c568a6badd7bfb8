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

}  // namespace npu_kernel
}  // namespace sglang

TORCH_LIBRARY_FRAGMENT(npu, m)
{
    m.def("sgl_kernel_npu_version() -> str", &sglang::npu_kernel::sgl_kernel_npu_version);
    m.def("decode_mla(Tensor q, Tensor k_nope_buffer, Tensor k_rope_buffer, Tensor(a!) att_out, Tensor kv_seq_lens, "
          "float sm_scale, int page_size, Tensor block_table, int num_splits=0) -> ()");
}

TORCH_LIBRARY_IMPL(npu, CUDA, m)
{
    m.impl("decode_mla", TORCH_FN(sglang::npu_kernel::decode_mla));
}
