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

namespace mi_sgl {

template <bool BF16>
__device__ __forceinline__ float ldh(uint16_t bits)
{
    if constexpr (BF16) return __uint_as_float((uint32_t)bits << 16);
    else return (float)__builtin_bit_cast(_Float16, bits);
}
template <bool BF16>
__device__ __forceinline__ uint16_t sth(float f)
{
    if constexpr (BF16) {
        uint32_t x = __float_as_uint(f);
        if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
        return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16);
    } else {
        // the fp32 value is rounded to fp16 as a SEPARATE step (the golden materialises the fp32 product first): keep the
        // compiler from folding the producing multiply into a mixed-precision v_fma_mixlo_f16, which rounds only once
        asm volatile("" : "+v"(f));
        return __builtin_bit_cast(uint16_t, (_Float16)f);
    }
}
// quant_per_tensor of the golden (:77-83): fp32 divide + add, round to fp16, clamp, round half to even, int8
__device__ __forceinline__ int8_t quant_pt(float x, float scale, float zp)
{
    float v = (float)(_Float16)(x / scale + zp);
    v = fminf(fmaxf(v, -128.f), 127.f);
    return (int8_t)(int)rintf(v);
}
constexpr int kMidThreads = 1024;      // one workgroup per token: 16 waves keep enough loads in flight to sum the split-K partials
__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < kMidThreads / 64; ++w) s += red[w];      // same order in every thread
    return s;
}

template <bool BF16>
__global__ __launch_bounds__(256) void pre_quant_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ scale_p,
                                                       const int8_t *__restrict__ zp_p, long long n, int8_t *__restrict__ out)
{
    const float scale = ldh<BF16>(scale_p[0]), zp = (float)zp_p[0];
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const uint4 v = *(const uint4 *)(x + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float f = ldh<BF16>((uint16_t)(w[j >> 1] >> (16 * (j & 1))));
        o[j >> 2] |= ((uint32_t)(uint8_t)quant_pt(f, scale, zp)) << (8 * (j & 3));
    }
    *(uint2 *)(out + i) = uint2{o[0], o[1]};
}

// quant_mode "per_token_quant_symm" (the reference's default, mla_preprocess_mix_bf16.hpp:437-483): scale = max|y| / 127 per row,
// q = int8(rint(clamp(fp16(y * (1 / scale))))), the row's scale is kept for the dequant of the following GEMM
__device__ __forceinline__ int8_t quant_tok(float y, float inv_scale)
{
    float v = y * inv_scale;
    asm volatile("" : "+v"(v));                          // fp32 product first, then a separate rounding to fp16 (no v_fma_mixlo)
    v = (float)(_Float16)v;
    v = fminf(fmaxf(v, -128.f), 127.f);
    return (int8_t)(int)rintf(v);
}
__device__ __forceinline__ float block_max(float v, float *red)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
    return m;
}

// one workgroup per token: row maximum, then the quantisation (the row is re-read from L2)
template <bool BF16>
__global__ __launch_bounds__(256) void pre_quant_token_kernel(const uint16_t *__restrict__ x, int H, int8_t *__restrict__ out,
                                                             float *__restrict__ tok_scale)
{
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    const uint16_t *row = x + (long long)n * H;
    float amax = 0.f;
    for (int i = tid * 8; i < H; i += 256 * 8) {
        const uint4 v = *(const uint4 *)(row + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(ldh<BF16>((uint16_t)(w[j >> 1] >> (16 * (j & 1))))));
    }
    amax = block_max(amax, red);
    const float scale = amax / 127.0f;
    const float inv = scale > 0.f ? 1.0f / scale : 0.f;       // an all-zero row quantises to zeros with scale 0
    if (tid == 0) tok_scale[n] = scale;
    for (int i = tid * 8; i < H; i += 256 * 8) {
        const uint4 v = *(const uint4 *)(row + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j >> 2] |= ((uint32_t)(uint8_t)quant_tok(ldh<BF16>((uint16_t)(w[j >> 1] >> (16 * (j & 1)))), inv)) << (8 * (j & 3));
        *(uint2 *)(out + (long long)n * H + i) = uint2{o[0], o[1]};
    }
}

constexpr int kKN = 512, kKR = 64, kQ = 1536, kMid = kKN + kKR + kQ;     // 2112

template <bool BF16>
__global__ __launch_bounds__(kMidThreads) void pre_mid_kernel(const int32_t *__restrict__ c1, int nparts, int ntok, const int32_t *__restrict__ bias0,
                                                     const float *__restrict__ descale0, const uint16_t *__restrict__ gamma1,
                                                     const uint16_t *__restrict__ beta1, const uint16_t *__restrict__ gamma2,
                                                     const uint16_t *__restrict__ cosv, const uint16_t *__restrict__ sinv,
                                                     const int32_t *__restrict__ slotmapping, const uint16_t *__restrict__ qscale1_p,
                                                     const int8_t *__restrict__ qoff1_p, float eps, int8_t *__restrict__ q8, uint16_t *__restrict__ kv_cache,
                                                     uint16_t *__restrict__ kv_cache_rope, const float *__restrict__ tok_scale_in,
                                                     float *__restrict__ tok_scale_out, int cache_mode, int block_size,
                                                     const uint16_t *__restrict__ ctkv_scale)
{
    __shared__ float f[kMid];
    __shared__ float red[kMidThreads / 64];
    const int n = blockIdx.x, tid = threadIdx.x;
    // per_token_quant_symm (tok_scale_in / tok_scale_out non-null): GEMM1 is dequantised with the token's own scale and no bias
    // (mla_preprocess_mix_bf16.hpp:389-421), the normalised q is requantised against its own row maximum (:437-483)
    const bool per_token = tok_scale_in != nullptr;
    const float qscale1 = per_token ? 1.f : ldh<BF16>(qscale1_p[0]), qoff1 = per_token ? 0.f : (float)qoff1_p[0];
    const float ts_in = per_token ? tok_scale_in[n] : 1.f;
    const int32_t *row = c1 + (long long)n * kMid;
    const long long part_stride = (long long)ntok * kMid;       // split-K partial products of GEMM1: exact int32 sum
    // Column j = tid, tid + 1024 and (threads 0..63) 2048 + tid: all three are summed TOGETHER, eight partials per step -- the loads of a
    // step are independent and unconditional (index clamped to the last partial, surplus values dropped), so the 14 partials cost two
    // memory round trips per thread.  (Column after column it was six: three passes of two steps, the third for 64 columns.)
    static_assert(kMid > 2 * kMidThreads && kMid <= 2 * kMidThreads + 64, "column plan of pre_mid");
    {
        const int j0 = tid, j1 = tid + kMidThreads, j2 = min(2 * kMidThreads + tid, kMid - 1);
        const bool has2 = tid < kMid - 2 * kMidThreads;
        const bool use_bias = bias0 && !per_token;
        int32_t acc0 = use_bias ? bias0[j0] : 0, acc1 = use_bias ? bias0[j1] : 0, acc2 = use_bias ? bias0[j2] : 0;
        const float d0 = descale0[j0], d1 = descale0[j1], d2 = descale0[j2];
        for (int p = 0; p < nparts; p += 8) {
            int32_t v0[8], v1[8], v2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int32_t *pr = row + (long long)min(p + u, nparts - 1) * part_stride;
                v0[u] = pr[j0];
                v1[u] = pr[j1];
                v2[u] = has2 ? pr[j2] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool live = p + u < nparts;
                acc0 += live ? v0[u] : 0;
                acc1 += live ? v1[u] : 0;
                acc2 += live ? v2[u] : 0;
            }
        }
        // the GEMM output is materialised in the I/O dtype (golden :95-107)
        float y0 = (float)acc0 * d0, y1 = (float)acc1 * d1, y2 = (float)acc2 * d2;
        if (per_token) y0 = y0 * ts_in, y1 = y1 * ts_in, y2 = y2 * ts_in;
        f[j0] = ldh<BF16>(sth<BF16>(y0));
        f[j1] = ldh<BF16>(sth<BF16>(y1));
        if (has2) f[j2] = ldh<BF16>(sth<BF16>(y2));
    }
    __syncthreads();
    // k_nope: RMSNorm * gamma2 -> cache
    float ss = 0.f;
    for (int j = tid; j < kKN; j += kMidThreads) ss += f[j] * f[j];
    const float rk = rsqrtf(block_sum(ss, red) / (float)kKN + eps);
    const long long slot = slotmapping[n];
    // Cache layouts (csrc/mla_preprocess/op_host/mla_preprocess.cpp:605-606; element positions per the reference test's
    // extract_from_nzcache, tests/python/sgl_kernel_npu/test_mla_preprocess.py:122-136):
    //   1 krope_ctkv    [slot][dim]
    //   3 nzcache       per block of block_size slots: [dim / 16][slot in block][16]
    //   2 int8_nzcache  k_nope as int8 = round(clamp(fp16(k_nope / ctkv_scale))) in [dim / 32][slot in block][32]; k_pe as mode 3
    const long long blk = cache_mode == 1 ? 0 : slot / block_size, inner = cache_mode == 1 ? 0 : slot % block_size;
    auto nz = [&](int dim, int j, int c0) { return blk * block_size * dim + ((long long)(j / c0) * block_size + inner) * c0 + j % c0; };
    if (cache_mode == 2) {
        const float cs = ldh<BF16>(ctkv_scale[0]);
        int8_t *kv8 = (int8_t *)kv_cache;
        for (int j = tid; j < kKN; j += kMidThreads) {
            const float y = (f[j] * rk) * ldh<BF16>(gamma2[j]);
            // quant_per_tensor of the golden (:74-80): fp32 divide, one rounding to fp16, clamp, round half to even
            float qv = y / cs;
            asm volatile("" : "+v"(qv));
            float h = (float)(_Float16)qv;
            h = fminf(fmaxf(h, -128.f), 127.f);
            kv8[nz(kKN, j, 32)] = (int8_t)(int)rintf(h);
        }
    } else {
        for (int j = tid; j < kKN; j += kMidThreads)
            kv_cache[cache_mode == 1 ? slot * kKN + j : nz(kKN, j, 16)] = sth<BF16>((f[j] * rk) * ldh<BF16>(gamma2[j]));
    }
    // k_pe: rotate-half RoPE -> rope cache
    if (tid < kKR) {
        const float x = f[kKN + tid];
        const float rot = tid < kKR / 2 ? -f[kKN + tid + kKR / 2] : f[kKN + tid - kKR / 2];
        const float c = ldh<BF16>(cosv[(long long)n * kKR + tid]), s = ldh<BF16>(sinv[(long long)n * kKR + tid]);
        kv_cache_rope[cache_mode == 1 ? slot * kKR + tid : nz(kKR, tid, 16)] = sth<BF16>(x * c + rot * s);
    }
    // q: RMSNorm * gamma1 + beta1 -> per-tensor INT8
    ss = 0.f;
    for (int j = tid; j < kQ; j += kMidThreads) ss += f[kKN + kKR + j] * f[kKN + kKR + j];
    const float rq = rsqrtf(block_sum(ss, red) / (float)kQ + eps);
    if (!per_token) {
        for (int j = tid; j < kQ; j += kMidThreads) {
            const float y = (f[kKN + kKR + j] * rq) * ldh<BF16>(gamma1[j]) + ldh<BF16>(beta1[j]);
            q8[(long long)n * kQ + j] = quant_pt(y, qscale1, qoff1);
        }
        return;
    }
    float amax = 0.f;
    for (int j = tid; j < kQ; j += kMidThreads)
        amax = fmaxf(amax, fabsf((f[kKN + kKR + j] * rq) * ldh<BF16>(gamma1[j]) + ldh<BF16>(beta1[j])));
    amax = block_max(amax, red);
    const float scale = amax / 127.0f;
    const float inv = scale > 0.f ? 1.0f / scale : 0.f;
    if (tid == 0) tok_scale_out[n] = scale;
    for (int j = tid; j < kQ; j += kMidThreads) {
        const float y = (f[kKN + kKR + j] * rq) * ldh<BF16>(gamma1[j]) + ldh<BF16>(beta1[j]);
        q8[(long long)n * kQ + j] = quant_tok(y, inv);
    }
}

}  // namespace mi_sgl

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
