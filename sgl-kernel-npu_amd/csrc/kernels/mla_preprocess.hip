// mla_preprocess elementwise / normalisation stages for gfx950 (the GEMMs are in mla_gemm.hip).
// Reference: csrc/mla_preprocess/op_host/mla_preprocess.cpp:623-704 + op_kernel/mla_preprocess_mix_bf16.hpp (AscendC MIX kernel);
// arithmetic pinned by the test golden golden2_pytorch (tests/python/sgl_kernel_npu/test_mla_preprocess.py:407-483):
//   q8  = int8(round(clamp(fp16(h / qscale0 + qoff0))))                                   -> mi_mla_pre_quant
//   f   = bf16((i32 GEMM1 + bias0) * descale0), split [512 k_nope | 64 k_pe | 1536 q]
//   k_nope = bf16(rms_norm(k_nope) * gamma2) -> kv_cache[slot];  k_pe = rope_half(k_pe) -> kv_cache_rope[slot]
//   q8' = int8(round(clamp(fp16((rms_norm(q) * gamma1 + beta1) / qscale1 + qoff1))))        -> mi_mla_pre_mid
//   qo  = bf16((i32 GEMM2 + bias1) * descale1) per head [128 nope | 64 pe]; q_pe = rope_half -> q_out1 -> mla_gemm.hip
// At decode sizes the op is weight-bandwidth bound (15 MB + 1536*Hq*192 B of INT8 weights + the bf16 wuk).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "mi_sgl_kernels.h"
#include "mla_pre_dev.h"

using namespace mi_sgl;

extern "C" int mi_mla_pre_quant(const void *x, const void *scale, const int8_t *zero_point, int64_t numel, int dtype, int8_t *out,
                                void *stream)
{
    if (numel < 0 || numel % 8 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16)) return MI_SGL_EINVAL;
    if (numel == 0) return MI_SGL_OK;
    if (!x || !out || !scale || !zero_point) return MI_SGL_EINVAL;
    const int blocks = (int)((numel / 8 + 255) / 256);
    if (dtype == MI_DTYPE_BF16)
        pre_quant_kernel<true><<<blocks, 256, 0, (hipStream_t)stream>>>((const uint16_t *)x, (const uint16_t *)scale, zero_point, numel, out);
    else
        pre_quant_kernel<false><<<blocks, 256, 0, (hipStream_t)stream>>>((const uint16_t *)x, (const uint16_t *)scale, zero_point, numel, out);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_mla_pre_quant_token(const void *x, int tokens, int hidden, int dtype, int8_t *out, float *tok_scale, void *stream)
{
    if (tokens < 0 || hidden <= 0 || hidden % 8 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16)) return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!x || !out || !tok_scale) return MI_SGL_EINVAL;
    if (dtype == MI_DTYPE_BF16) pre_quant_token_kernel<true><<<tokens, 256, 0, (hipStream_t)stream>>>((const uint16_t *)x, hidden, out, tok_scale);
    else pre_quant_token_kernel<false><<<tokens, 256, 0, (hipStream_t)stream>>>((const uint16_t *)x, hidden, out, tok_scale);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_mla_pre_mid(const int32_t *gemm1_i32, int num_partials, const int32_t *bias0, const float *descale0, const void *gamma1,
                              const void *beta1, const void *gamma2, const void *cos, const void *sin, const int32_t *slotmapping,
                              const void *quant_scale1, const int8_t *quant_offset1, float eps, int tokens, int dtype, int8_t *q_int8,
                              void *kv_cache, void *kv_cache_rope, const float *tok_scale_in, float *tok_scale_out, int cache_mode,
                              int block_size, const void *ctkv_scale, void *stream)
{
    if ((tok_scale_in == nullptr) != (tok_scale_out == nullptr)) return MI_SGL_EINVAL;
    if (cache_mode < 1 || cache_mode > 3 || (cache_mode != 1 && block_size <= 0) || (cache_mode == 2 && !ctkv_scale)) return MI_SGL_EINVAL;
    if (tokens < 0 || num_partials < 1 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) ||
        (!tok_scale_in && (!quant_scale1 || !quant_offset1)))
        return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!gemm1_i32 || !descale0 || !gamma1 || !beta1 || !gamma2 || !cos || !sin || !slotmapping || !q_int8 || !kv_cache || !kv_cache_rope)
        return MI_SGL_EINVAL;
#define MI_MID(B)                                                                                                                  \
    pre_mid_kernel<B><<<tokens, kMidThreads, 0, (hipStream_t)stream>>>(gemm1_i32, num_partials, tokens, bias0, descale0, (const uint16_t *)gamma1,                 \
                                                               (const uint16_t *)beta1, (const uint16_t *)gamma2, (const uint16_t *)cos, \
                                                               (const uint16_t *)sin, slotmapping, (const uint16_t *)quant_scale1, quant_offset1, eps,  \
                                                               q_int8, (uint16_t *)kv_cache, (uint16_t *)kv_cache_rope, tok_scale_in, tok_scale_out, \
                                                               cache_mode, block_size, (const uint16_t *)ctkv_scale)
    if (dtype == MI_DTYPE_BF16) MI_MID(true); else MI_MID(false);
#undef MI_MID
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}
