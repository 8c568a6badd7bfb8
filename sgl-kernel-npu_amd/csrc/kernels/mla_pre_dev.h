// Device-side stages of mla_preprocess shared by mla_preprocess.hip (the stand-alone launches) and mla_gemm.hip (the one-launch form):
// quantisation of the hidden states, and the "middle" stage between the two GEMMs.  See mla_preprocess.hip for the arithmetic contract.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

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
// The stages below are written for kMidThreads "logical threads" per token and run on THREADS hardware threads, every one playing the
// logical threads tid, tid + THREADS, ...: the stand-alone launch has THREADS = 1024, the one-launch form of mla_gemm.hip 512.  Sums and
// maxima are formed per logical thread, reduced inside the logical thread's wave (same lanes) and added over the 16 logical waves in the
// same order -- so both forms produce the same bits.
template <int THREADS>
__device__ __forceinline__ float block_sum(const float (&v)[kMidThreads / THREADS], float *red)
{
    constexpr int V = kMidThreads / THREADS;
    float w[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        w[i] = v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) w[i] += __shfl_xor(w[i], off, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < V; ++i) red[(threadIdx.x + i * THREADS) >> 6] = w[i];
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kMidThreads / 64; ++k) s += red[k];      // same order in every thread
    return s;
}
// `i8` = index of this thread's group of eight elements; the caller strides it over the tensor
template <bool BF16, bool SC1>
__device__ __forceinline__ void pre_quant_eight_loaded(uint4 v, float scale, float zp, long long i, int8_t *__restrict__ out)
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float f = ldh<BF16>((uint16_t)(w[j >> 1] >> (16 * (j & 1))));
        o[j >> 2] |= ((uint32_t)(uint8_t)quant_pt(f, scale, zp)) << (8 * (j & 3));
    }
    // SC1: a write-through (device-scope) store, for consumers in the same launch on other XCDs (mla_gemm.hip, one-launch form)
    if constexpr (SC1) __hip_atomic_store((unsigned long long *)(out + i), (unsigned long long)o[0] | ((unsigned long long)o[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(uint2 *)(out + i) = uint2{o[0], o[1]};
}
template <bool BF16, bool SC1>
__device__ __forceinline__ void pre_quant_eight(const uint16_t *__restrict__ x, float scale, float zp, long long i, int8_t *__restrict__ out)
{
    pre_quant_eight_loaded<BF16, SC1>(*(const uint4 *)(x + i), scale, zp, i, out);
}
template <bool BF16>
__global__ __launch_bounds__(256) void pre_quant_kernel(const uint16_t *__restrict__ x, const uint16_t *__restrict__ scale_p,
                                                       const int8_t *__restrict__ zp_p, long long n, int8_t *__restrict__ out)
{
    const float scale = ldh<BF16>(scale_p[0]), zp = (float)zp_p[0];
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    pre_quant_eight<BF16, false>(x, scale, zp, i, out);
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
__device__ __forceinline__ float block_max(float v, float *red)            // max is order-free: any thread count
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
template <bool BF16, int THREADS, bool SC1>
__device__ __forceinline__ void pre_quant_token_body(const uint16_t *__restrict__ x, int H, int8_t *__restrict__ out, float *__restrict__ tok_scale,
                                                     int n, float *red)
{
    const int tid = threadIdx.x;
    const uint16_t *row = x + (long long)n * H;
    float amax = 0.f;
    for (int i = tid * 8; i < H; i += THREADS * 8) {
        const uint4 v = *(const uint4 *)(row + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(ldh<BF16>((uint16_t)(w[j >> 1] >> (16 * (j & 1))))));
    }
    amax = block_max(amax, red);
    const float scale = amax / 127.0f;
    const float inv = scale > 0.f ? 1.0f / scale : 0.f;       // an all-zero row quantises to zeros with scale 0
    if (tid == 0) {
        if constexpr (SC1) __hip_atomic_store((uint32_t *)(tok_scale + n), __float_as_uint(scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else tok_scale[n] = scale;
    }
    for (int i = tid * 8; i < H; i += THREADS * 8) {
        const uint4 v = *(const uint4 *)(row + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            o[j >> 2] |= ((uint32_t)(uint8_t)quant_tok(ldh<BF16>((uint16_t)(w[j >> 1] >> (16 * (j & 1)))), inv)) << (8 * (j & 3));
        int8_t *dst = out + (long long)n * H + i;
        if constexpr (SC1) __hip_atomic_store((unsigned long long *)dst, (unsigned long long)o[0] | ((unsigned long long)o[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *(uint2 *)dst = uint2{o[0], o[1]};
    }
}
template <bool BF16>
__global__ __launch_bounds__(256) void pre_quant_token_kernel(const uint16_t *__restrict__ x, int H, int8_t *__restrict__ out,
                                                             float *__restrict__ tok_scale)
{
    __shared__ float red[4];
    pre_quant_token_body<BF16, 256, false>(x, H, out, tok_scale, (int)blockIdx.x, red);
}

constexpr int kKN = 512, kKR = 64, kQ = 1536, kMid = kKN + kKR + kQ;     // 2112

// One token row: n.  `f` [kMid] and `red` [kMidThreads / 64] are workgroup-shared scratch.  SC1: q8 / tok_scale_out leave through
// device-scope (write-through) stores for a consumer in the same launch.
template <bool BF16, int THREADS, bool SC1, int PB = 8 /* split-K partials requested per memory round trip */>
__device__ __forceinline__ void pre_mid_body(const int32_t *__restrict__ c1, int nparts, int ntok, const int32_t *__restrict__ bias0,
                                             const float *__restrict__ descale0, const uint16_t *__restrict__ gamma1,
                                             const uint16_t *__restrict__ beta1, const uint16_t *__restrict__ gamma2,
                                             const uint16_t *__restrict__ cosv, const uint16_t *__restrict__ sinv,
                                             const int32_t *__restrict__ slotmapping, const uint16_t *__restrict__ qscale1_p,
                                             const int8_t *__restrict__ qoff1_p, float eps, int8_t *__restrict__ q8, uint16_t *__restrict__ kv_cache,
                                             uint16_t *__restrict__ kv_cache_rope, const float *__restrict__ tok_scale_in,
                                             float *__restrict__ tok_scale_out, int cache_mode, int block_size,
                                             const uint16_t *__restrict__ ctkv_scale, int n, float *f, float *red)
{
    static_assert(kMidThreads % THREADS == 0, "logical threads per hardware thread");
    constexpr int V = kMidThreads / THREADS;
    const int tid = threadIdx.x;
    // per_token_quant_symm (tok_scale_in / tok_scale_out non-null): GEMM1 is dequantised with the token's own scale and no bias
    // (mla_preprocess_mix_bf16.hpp:389-421), the normalised q is requantised against its own row maximum (:437-483)
    const bool per_token = tok_scale_in != nullptr;
    const float qscale1 = per_token ? 1.f : ldh<BF16>(qscale1_p[0]), qoff1 = per_token ? 0.f : (float)qoff1_p[0];
    const float ts_in = per_token ? tok_scale_in[n] : 1.f;
    const int32_t *row = c1 + (long long)n * kMid;
    const long long part_stride = (long long)ntok * kMid;       // split-K partial products of GEMM1: exact int32 sum
    // Column j = T, T + 1024 and (logical threads 0..63) 2048 + T: all three are summed TOGETHER, eight partials per step -- the loads of a
    // step are independent and unconditional (index clamped to the last partial, surplus values dropped), so the 14 partials cost two
    // memory round trips per thread.  (Column after column it was six: three passes of two steps, the third for 64 columns.)
    static_assert(kMid > 2 * kMidThreads && kMid <= 2 * kMidThreads + 64, "column plan of pre_mid");
    {
        const bool use_bias = bias0 && !per_token;
        int j0[V], j1[V], j2[V];
        bool has2[V];
        int32_t acc0[V], acc1[V], acc2[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int T = tid + v * THREADS;
            j0[v] = T, j1[v] = T + kMidThreads, j2[v] = min(2 * kMidThreads + T, kMid - 1);
            has2[v] = T < kMid - 2 * kMidThreads;
            acc0[v] = use_bias ? bias0[j0[v]] : 0, acc1[v] = use_bias ? bias0[j1[v]] : 0, acc2[v] = use_bias ? bias0[j2[v]] : 0;
        }
        for (int p = 0; p < nparts; p += PB) {
            int32_t v0[V][PB], v1[V][PB], v2[V][PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int32_t *pr = row + (long long)min(p + u, nparts - 1) * part_stride;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    v0[v][u] = pr[j0[v]];
                    v1[v][u] = pr[j1[v]];
                    v2[v][u] = has2[v] ? pr[j2[v]] : 0;
                }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const bool live = p + u < nparts;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    acc0[v] += live ? v0[v][u] : 0;
                    acc1[v] += live ? v1[v][u] : 0;
                    acc2[v] += live ? v2[v][u] : 0;
                }
            }
        }
        // the GEMM output is materialised in the I/O dtype (golden :95-107)
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float y0 = (float)acc0[v] * descale0[j0[v]], y1 = (float)acc1[v] * descale0[j1[v]], y2 = (float)acc2[v] * descale0[j2[v]];
            if (per_token) y0 = y0 * ts_in, y1 = y1 * ts_in, y2 = y2 * ts_in;
            f[j0[v]] = ldh<BF16>(sth<BF16>(y0));
            f[j1[v]] = ldh<BF16>(sth<BF16>(y1));
            if (has2[v]) f[j2[v]] = ldh<BF16>(sth<BF16>(y2));
        }
    }
    __syncthreads();
    // k_nope: RMSNorm * gamma2 -> cache.  (Logical thread T adds the squares of columns T, T + 1024, ...: kept so for the summation order.)
    float ssv[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        ssv[v] = 0.f;
        for (int j = tid + v * THREADS; j < kKN; j += kMidThreads) ssv[v] += f[j] * f[j];
    }
    const float rk = rsqrtf(block_sum<THREADS>(ssv, red) / (float)kKN + eps);
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
        for (int j = tid; j < kKN; j += THREADS) {
            const float y = (f[j] * rk) * ldh<BF16>(gamma2[j]);
            // quant_per_tensor of the golden (:74-80): fp32 divide, one rounding to fp16, clamp, round half to even
            float qv = y / cs;
            asm volatile("" : "+v"(qv));
            float h = (float)(_Float16)qv;
            h = fminf(fmaxf(h, -128.f), 127.f);
            kv8[nz(kKN, j, 32)] = (int8_t)(int)rintf(h);
        }
    } else {
        for (int j = tid; j < kKN; j += THREADS)
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
#pragma unroll
    for (int v = 0; v < V; ++v) {
        ssv[v] = 0.f;
        for (int j = tid + v * THREADS; j < kQ; j += kMidThreads) ssv[v] += f[kKN + kKR + j] * f[kKN + kKR + j];
    }
    const float rq = rsqrtf(block_sum<THREADS>(ssv, red) / (float)kQ + eps);
    auto qnorm = [&](int j) { return (f[kKN + kKR + j] * rq) * ldh<BF16>(gamma1[j]) + ldh<BF16>(beta1[j]); };
    float inv = 0.f;
    if (per_token) {
        float amax = 0.f;
        for (int j = tid; j < kQ; j += THREADS) amax = fmaxf(amax, fabsf(qnorm(j)));
        amax = block_max(amax, red);
        const float scale = amax / 127.0f;
        inv = scale > 0.f ? 1.0f / scale : 0.f;
        if (tid == 0) {
            if constexpr (SC1) __hip_atomic_store((uint32_t *)(tok_scale_out + n), __float_as_uint(scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else tok_scale_out[n] = scale;
        }
    }
    auto qone = [&](int j) -> int8_t { return per_token ? quant_tok(qnorm(j), inv) : quant_pt(qnorm(j), qscale1, qoff1); };
    if constexpr (SC1) {
        for (int j = tid * 4; j < kQ; j += THREADS * 4) {     // four columns per thread: one 4-byte write-through store
            const uint32_t w = (uint32_t)(uint8_t)qone(j) | ((uint32_t)(uint8_t)qone(j + 1) << 8) | ((uint32_t)(uint8_t)qone(j + 2) << 16) |
                               ((uint32_t)(uint8_t)qone(j + 3) << 24);
            __hip_atomic_store((uint32_t *)(q8 + (long long)n * kQ + j), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        for (int j = tid; j < kQ; j += THREADS) q8[(long long)n * kQ + j] = qone(j);
    }
}

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
    pre_mid_body<BF16, kMidThreads, false>(c1, nparts, ntok, bias0, descale0, gamma1, beta1, gamma2, cosv, sinv, slotmapping, qscale1_p, qoff1_p, eps, q8,
                                           kv_cache, kv_cache_rope, tok_scale_in, tok_scale_out, cache_mode, block_size, ctkv_scale, (int)blockIdx.x, f, red);
}


}  // namespace mi_sgl
