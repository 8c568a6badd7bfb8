// Row statistics and row scalings for gfx950, HBM-bound: L1 normalisation, RMSNorm without a weight, and the two halves of a split
// RMSNorm (row variance; x * rsqrt(variance + eps) * weight).  Replace the reference's Triton-Ascend kernels:
//   l1_norm                        python/sgl_kernel_npu/sgl_kernel_npu/norm/l1_norm.py:7-38
//   fused_rmsnorm_without_weight   python/sgl_kernel_npu/sgl_kernel_npu/norm/rmsnorm_without_weight.py:30-76
//   fused_variance                 python/sgl_kernel_npu/sgl_kernel_npu/norm/rmsnorm_split.py:124-161
//   fused_rsqrt_mul                python/sgl_kernel_npu/sgl_kernel_npu/norm/rmsnorm_split.py:34-97
//   fused_scale_shift              python/sgl_kernel_npu/sgl_kernel_npu/norm/scale_shift.py:9-183
//   split_qkv_tp_rmsnorm_rope      python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_tp_rmsnorm_rope.py:7-288
//   fused_split_qk_norm            python/sgl_kernel_npu/sgl_kernel_npu/norm/fused_split_qk_norm.py:6-134
//   swiglu_oai                     python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_oai.py:7-104
//   swiglu_oai_quant               python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_oai_quant.py:39-211
//   situ / situ_and_mul(_quant)    python/sgl_kernel_npu/sgl_kernel_npu/activation/situ.py:11-480
//   mix_fused (attn residual)      python/sgl_kernel_npu/sgl_kernel_npu/kimi_k3/attn_residual.py:7-111
//   mul_add                        python/sgl_kernel_npu/sgl_kernel_npu/moe/mul_add.py:9-60
//   zero_experts_compute_identity  python/sgl_kernel_npu/sgl_kernel_npu/moe/zero_experts_compute_identity.py:6-81
// The reference tests run them on fp32 tensors (tests/python/sgl_kernel_npu/test_{l1_norm,rmsnorm_without_weight,rmsnorm_split}.py), models
// on bf16 / fp16: all three element types, arithmetic in fp32 throughout.
// MI355X design: one wave64 per row, 16-byte loads; a row of up to 8192 16-bit / 4096 fp32 elements stays in registers between the reduction and the
// scaling (one pass over HBM), longer rows are read a second time (out of L2).  L1 rows of at most 32 elements (a router's 8 expert
// scores per token: 2048 x 8 in the reference test) take one lane each.
// Algorithmic bytes: rows x cols x (in + out element sizes) (+ 4 rows for the variance).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_sgl_kernels.h"

namespace mi_sgl {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// element access by dtype code (wave-uniform): 0 = bf16, 1 = fp16, 2 = fp32
template <int DT> struct Elem;
template <> struct Elem<MI_DTYPE_BF16> {
    typedef uint16_t T;
    static constexpr int kPer16 = 8;
    static __device__ __forceinline__ float ld(T v) { return __uint_as_float((uint32_t)v << 16); }
    static __device__ __forceinline__ T st(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
};
template <> struct Elem<MI_DTYPE_F16> {
    typedef uint16_t T;
    static constexpr int kPer16 = 8;
    static __device__ __forceinline__ float ld(T v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ T st(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
};
template <> struct Elem<MI_DTYPE_F32> {
    typedef float T;
    static constexpr int kPer16 = 4;
    static __device__ __forceinline__ float ld(T v) { return v; }
    static __device__ __forceinline__ T st(float f) { return f; }
};

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// what a row kernel computes
enum { kOpL1 = 0, kOpRms = 1, kOpVariance = 2, kOpRsqrtMul = 3 };

// 16 bytes of a row -> fp32 values (n = elements per 16 bytes)
template <int DT>
__device__ __forceinline__ void load16(const typename Elem<DT>::T *p, float *f)
{
    const u32x4 v = *(const u32x4 *)p;
    if constexpr (DT == MI_DTYPE_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __uint_as_float(v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = Elem<DT>::ld((uint16_t)(v[j] & 0xFFFFu));
            f[2 * j + 1] = Elem<DT>::ld((uint16_t)(v[j] >> 16));
        }
    }
}
// 16 bytes already in registers -> fp32 values
template <int DT>
__device__ __forceinline__ void unpack16(const u32x4 v, float *f)
{
    if constexpr (DT == MI_DTYPE_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = __uint_as_float(v[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f[2 * j] = Elem<DT>::ld((uint16_t)(v[j] & 0xFFFFu));
            f[2 * j + 1] = Elem<DT>::ld((uint16_t)(v[j] >> 16));
        }
    }
}
template <int DT>
__device__ __forceinline__ void store16(typename Elem<DT>::T *p, const float *f)
{
    u32x4 v;
    if constexpr (DT == MI_DTYPE_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = __float_as_uint(f[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (uint32_t)Elem<DT>::st(f[2 * j]) | ((uint32_t)Elem<DT>::st(f[2 * j + 1]) << 16);
    }
    *(u32x4 *)p = v;
}

// One wave per row.  DT = element type of x (and of the output, except L1: fp32 out; variance: DT out).  kChunks 16-byte pieces per lane
// are kept in registers when the row fits (cols <= 64 * kChunks * kPer16), otherwise the row is read twice.
template <int OP, int DT>
__global__ __launch_bounds__(256) void row_kernel(const typename Elem<DT>::T *__restrict__ x, long long rows, int cols, float eps,
                                                  const typename Elem<DT>::T *__restrict__ variance, const typename Elem<DT>::T *__restrict__ weight,
                                                  void *__restrict__ out_v)
{
    typedef typename Elem<DT>::T T;
    constexpr int N = Elem<DT>::kPer16;
    constexpr int kChunks = 16;                             // rows of up to 8192 (16-bit) / 4096 (fp32) elements stay in registers
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T *xr = x + row * (long long)cols;
    const bool vec = (cols % N) == 0;                       // rows then start 16-byte aligned (the base pointer is)
    const bool in_regs = vec && cols <= 64 * kChunks * N;
    float keep[kChunks][N];
    float acc = 0.f;
    if (OP != kOpRsqrtMul) {
        if (in_regs) {
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int i = (c * 64 + lane) * N;
                if (i < cols) {
                    load16<DT>(xr + i, keep[c]);
#pragma unroll
                    for (int e = 0; e < N; ++e) acc += (OP == kOpL1) ? keep[c][e] : keep[c][e] * keep[c][e];
                }
            }
        } else if (vec) {
            for (int i = lane * N; i < cols; i += 64 * N) {
                float f[N];
                load16<DT>(xr + i, f);
#pragma unroll
                for (int e = 0; e < N; ++e) acc += (OP == kOpL1) ? f[e] : f[e] * f[e];
            }
        } else {
            for (int i = lane; i < cols; i += 64) {
                const float f = Elem<DT>::ld(xr[i]);
                acc += (OP == kOpL1) ? f : f * f;
            }
        }
        acc = wave_sum_f(acc);
    }
    if (OP == kOpVariance) {                                // sum / hidden_size (rmsnorm_split.py:152-154)
        if (lane == 0) ((T *)out_v)[row] = Elem<DT>::st(acc / (float)cols);
        return;
    }
    // the row's factor: 1 / sum (l1_norm.py:23), rsqrt(sum * (1 / hidden) + eps) (rmsnorm_without_weight.py:52-54),
    // rsqrt(variance + eps) (rmsnorm_split.py:72)
    float factor;
    if (OP == kOpL1) factor = acc;                          // divided below, as the reference does
    else if (OP == kOpRms) factor = rsqrtf(acc * (1.0f / (float)cols) + eps);
    else factor = rsqrtf(Elem<DT>::ld(variance[row]) + eps);
    auto scale = [&](float v, int col) -> float {
        if (OP == kOpL1) return v / factor;
        if (OP == kOpRms) return v * factor;
        return (v * factor) * Elem<DT>::ld(weight[col]);
    };
    if (OP == kOpL1) {                                      // fp32 output whatever the input type (l1_norm.py:24-26, :32-34)
        float *o = (float *)out_v + row * (long long)cols;
        if (in_regs) {
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int i = (c * 64 + lane) * N;
                if (i < cols) {
#pragma unroll
                    for (int e = 0; e < N; ++e) o[i + e] = scale(keep[c][e], i + e);
                }
            }
        } else {
            for (int i = lane; i < cols; i += 64) o[i] = scale(Elem<DT>::ld(xr[i]), i);
        }
        return;
    }
    T *o = (T *)out_v + row * (long long)cols;
    if (OP == kOpRms && in_regs) {
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int i = (c * 64 + lane) * N;
            if (i < cols) {
                float f[N];
#pragma unroll
                for (int e = 0; e < N; ++e) f[e] = scale(keep[c][e], i + e);
                store16<DT>(o + i, f);
            }
        }
    } else if (vec) {
        for (int i = lane * N; i < cols; i += 64 * N) {
            float f[N];
            load16<DT>(xr + i, f);
#pragma unroll
            for (int e = 0; e < N; ++e) f[e] = scale(f[e], i + e);
            store16<DT>(o + i, f);
        }
    } else {
        for (int i = lane; i < cols; i += 64) o[i] = Elem<DT>::st(scale(Elem<DT>::ld(xr[i]), i));
    }
}

// rows of at most 32 elements (a router's expert scores): one lane per row
template <int DT>
__global__ __launch_bounds__(256) void l1_small_kernel(const typename Elem<DT>::T *__restrict__ x, long long rows, int cols, float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const typename Elem<DT>::T *xr = x + row * (long long)cols;
    float v[32];
    float s = 0.f;
    for (int i = 0; i < cols; ++i) {
        v[i] = Elem<DT>::ld(xr[i]);
        s += v[i];
    }
    for (int i = 0; i < cols; ++i) out[row * (long long)cols + i] = v[i] / s;
}

template <int OP>
int launch_rows(const void *x, long long rows, int cols, float eps, const void *variance, const void *weight, int dtype, void *out, void *stream)
{
    if (rows < 0 || cols <= 0 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16 && dtype != MI_DTYPE_F32)) return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!x || !out || (OP == kOpRsqrtMul && (!variance || !weight))) return MI_SGL_EINVAL;
    if (rows > (1ll << 32)) return MI_SGL_EINVAL;        // four rows per workgroup, grid.x < 2^31
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
#define MI_ROW(DT) row_kernel<OP, DT><<<blocks, 256, 0, st>>>((const typename Elem<DT>::T *)x, rows, cols, eps, (const typename Elem<DT>::T *)variance, \
                                                             (const typename Elem<DT>::T *)weight, out)
    if (dtype == MI_DTYPE_BF16) MI_ROW(MI_DTYPE_BF16);
    else if (dtype == MI_DTYPE_F16) MI_ROW(MI_DTYPE_F16);
    else MI_ROW(MI_DTYPE_F32);
#undef MI_ROW
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

// [q_lora | kv_lora | rope] row of the MLA down-projection -> RMSNorm(q_lora) * w (+ b), RMSNorm(kv_lora) * w (+ b), rope part copied
// (norm/fused_split_qk_norm.py:6-91): y = (x * rsqrt(sum(x^2) / n + eps)) * w (+ b) in fp32 (:39-47, :62-70).  Two waves per row: wave 0 the
// q part, wave 1 the kv part and the copy.
template <int DT>
__global__ __launch_bounds__(128) void split_qk_norm_kernel(const typename Elem<DT>::T *__restrict__ x, int q_rank, int kv_rank, int rope_dim, float eps,
                                                            const typename Elem<DT>::T *__restrict__ qw, const typename Elem<DT>::T *__restrict__ qb,
                                                            const typename Elem<DT>::T *__restrict__ kw, const typename Elem<DT>::T *__restrict__ kb,
                                                            typename Elem<DT>::T *__restrict__ q, typename Elem<DT>::T *__restrict__ k_nope,
                                                            typename Elem<DT>::T *__restrict__ k_pe)
{
    typedef typename Elem<DT>::T T;
    const long long row = blockIdx.x;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const T *xr = x + row * (long long)(q_rank + kv_rank + rope_dim);
    const int n = part ? kv_rank : q_rank;
    const T *src = part ? xr + q_rank : xr;
    const T *w = part ? kw : qw, *b = part ? kb : qb;
    T *dst = part ? k_nope + row * (long long)kv_rank : q + row * (long long)q_rank;
    float ss = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float f = Elem<DT>::ld(src[i]);
        ss += f * f;
    }
    const float rstd = rsqrtf(wave_sum_f(ss) / (float)n + eps);
    for (int i = lane; i < n; i += 64) {
        float y = (Elem<DT>::ld(src[i]) * rstd) * Elem<DT>::ld(w[i]);
        if (b) y = y + Elem<DT>::ld(b[i]);
        dst[i] = Elem<DT>::st(y);
    }
    if (part) {
        const T *pe = xr + q_rank + kv_rank;
        for (int i = lane; i < rope_dim; i += 64) k_pe[row * (long long)rope_dim + i] = pe[i];
    }
}

// GPT-OSS SwiGLU on INTERLEAVED gate / up columns (activation/swiglu_oai.py:7-50): gate = x[2j] clamped from above at limit, up = x[2j + 1]
// clamped to [-limit, limit], out[j] = (up + 1) * gate * (1 / (1 + exp(-gate * alpha))) (:36-41).  fp32 arithmetic, output in x's dtype.
template <int DT>
__global__ __launch_bounds__(256) void swiglu_oai_kernel(const typename Elem<DT>::T *__restrict__ x, long long n_out, float alpha, float limit,
                                                         typename Elem<DT>::T *__restrict__ out)
{
    constexpr int N = Elem<DT>::kPer16;                     // outputs per thread and step: 2 N inputs
    const long long stride = (long long)gridDim.x * 256 * N;
    for (long long o0 = ((long long)blockIdx.x * 256 + threadIdx.x) * N; o0 < n_out; o0 += stride) {
        float a[N], b[N], r[N];
        load16<DT>(x + 2 * o0, a);
        load16<DT>(x + 2 * o0 + N, b);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            float g = (e < N / 2) ? a[2 * e] : b[2 * e - N];
            float u = (e < N / 2) ? a[2 * e + 1] : b[2 * e + 1 - N];
            g = g > limit ? limit : g;
            u = u > limit ? limit : u;
            u = u < -limit ? -limit : u;
            const float sig = 1.0f / (1.0f + __expf(-g * alpha));
            r[e] = (u + 1.0f) * (g * sig);
        }
        store16<DT>(out + o0, r);
    }
}

// GPT-OSS SwiGLU on CONCATENATED [gate | up] halves with optional per-row INT8 (activation/swiglu_oai_quant.py:39-112): gate = min(x1, limit),
// up = clamp(x2, +-limit), out = gate * sigmoid(gate * alpha) * (up + 1) (:83-85); quantised: scale = max|out| / 127 (:89), q = int8 of
// (out / scale) rounded to the I/O dtype first (:96) and then converted with saturation (:97).  The reference leaves that conversion to the
// backend's cast; here it TRUNCATES toward zero, the float -> int conversion of the Triton language the reference kernel is written in
// (stated assumption: the reference holds no test or vector for this function).  Rows beyond the group list's total are left untouched.
// One wave per row, 16-byte loads; rows of up to 6144 outputs stay in registers between the maximum and the conversion, longer ones are
// recomputed from a second read.
// tanh to ~1e-6 relative in ~12 VALU operations (libm's tanhf costs ~40 and made SiTU VALU-bound at 2.3 TB/s): the odd polynomial
// x - x^3/3 + 2x^5/15 - 17x^7/315 below |x| = 0.25 (next term 8e-8 relative there), 1 - 2 / (exp(2|x|) + 1) above (no cancellation: the
// result is >= 0.24)
__device__ __forceinline__ float tanh_fast(float x)
{
    const float ax = fabsf(x), x2 = x * x;
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * ax) + 1.0f);
    const float p = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f - x2 * 0.05396825f)));
    return ax < 0.25f ? p : copysignf(t, x);
}

// ACT 1: SiTU (activation/situ.py:11-90, :361-427): gate' = beta * tanh(gate / beta) * sigmoid(gate), up' = linear_beta * tanh(up / linear_beta)
// (optional), out = gate' * up' (:61-64); quantised: scale = max(max|out| / 127, 1e-30) (:67), q = clamp(floor(out / scale + 0.5), -128, 127)
// (:78-80) -- the rounding is explicit here.  alpha = beta, limit = linear_beta (<= 0: the up path is left alone).
template <int DT, bool I64, int ACT = 0>
__global__ __launch_bounds__(256) void swiglu_oai_quant_kernel(const typename Elem<DT>::T *__restrict__ x, const void *__restrict__ group_list, int num_groups,
                                                               int group_list_type, long long rows, int half, float alpha, float limit, int need_quant,
                                                               void *__restrict__ out, float *__restrict__ scale)
{
    typedef typename Elem<DT>::T T;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    long long total = rows;
    if (group_list) {                                     // 0 = cumulative counts (last entry = total, :63-64), 1 = counts (:66-71)
        if (group_list_type == 0) {
            total = I64 ? ((const long long *)group_list)[num_groups - 1] : (long long)((const int *)group_list)[num_groups - 1];
        } else {
            long long sacc = 0;
            for (int i = lane; i < num_groups; i += 64) sacc += I64 ? ((const long long *)group_list)[i] : (long long)((const int *)group_list)[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off, 64);
            total = sacc;
        }
    }
    if (row >= rows || row >= total) return;
    const T *xr = x + row * 2 * (long long)half;
    const float inv_alpha = 1.0f / alpha, inv_limit = limit > 0.f ? 1.0f / limit : 0.f;
    auto act = [&](float g, float u) -> float {
        if constexpr (ACT == 1) {
            const float ga = (alpha * tanh_fast(g * inv_alpha)) * (1.0f / (1.0f + __expf(-g)));
            if (limit > 0.f) u = limit * tanh_fast(u * inv_limit);
            return ga * u;
        } else {
            g = fminf(g, limit);
            u = fminf(fmaxf(u, -limit), limit);
            return (g * (1.0f / (1.0f + __expf(-g * alpha)))) * (u + 1.0f);
        }
    };
    auto value = [&](int j) -> float { return act(Elem<DT>::ld(xr[j]), Elem<DT>::ld(xr[half + j])); };
    // rounded to the I/O dtype first (:96), then truncated and saturated (:97); 0 / 0 (an all-zero row) -> 0
    // v / sc for a row's one sc: through the correctly rounded reciprocal and one residual correction (q0 = v r, q = q0 + (v - sc q0) r --
    // the IEEE quotient whenever nothing under- or overflows, three operations instead of the ~12 of a division; sc == 0 keeps 0 / 0 = NaN)
    float rsc = 0.f;
    auto div_sc = [&](float v, float sc) -> float {
        const float q0 = v * rsc;
        const float q = __builtin_fmaf(__builtin_fmaf(-sc, q0, v), rsc, q0);
        return (sc > 1e-30f && sc < 1e30f) ? q : v / sc;
    };
    auto quant = [&](float v, float sc) -> int {
        if constexpr (ACT == 1) return (int)fminf(fmaxf(floorf(div_sc(v, sc) + 0.5f), -128.f), 127.f);
        float r = Elem<DT>::ld(Elem<DT>::st(div_sc(v, sc)));
        r = r != r ? 0.f : truncf(r);
        return (int)fminf(fmaxf(r, -128.f), 127.f);
    };
    constexpr int N = Elem<DT>::kPer16;
    const bool vec = (half % N) == 0;                      // both halves of every row then start 16-byte aligned
    if (!need_quant) {
        T *o = (T *)out + row * (long long)half;
        if (vec) {
            // four 16-byte pieces of each half per lane and step, requested TOGETHER and unconditionally (index clamped into the row: a
            // conditional load gets a block and a wait of its own, which left one or two loads in flight per wave: 3.2 TB/s)
            for (int j0 = lane * N; j0 < half; j0 += 4 * 64 * N) {
                u32x4 rg[4], ru[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = min(j0 + b * 64 * N, half - N);
                    rg[b] = *(const u32x4 *)(xr + j);
                    ru[b] = *(const u32x4 *)(xr + half + j);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = j0 + b * 64 * N;
                    if (j >= half) break;
                    float g[N], u[N];
                    unpack16<DT>(rg[b], g);
                    unpack16<DT>(ru[b], u);
#pragma unroll
                    for (int e = 0; e < N; ++e) g[e] = act(g[e], u[e]);
                    store16<DT>(o + j, g);
                }
            }
        } else {
            for (int j = lane; j < half; j += 64) o[j] = Elem<DT>::st(value(j));
        }
        return;
    }
    float amax = 0.f;
    constexpr int kChunks = 12;                            // rows of up to 6144 outputs (the most SiTU quantises) stay in registers between the maximum and the conversion
    const bool in_regs = vec && half <= 64 * kChunks * N;
    float keep[kChunks][N];                                // the row's activations (the quantising forms are VALU-bound: computed once)
    if (in_regs) {
        // all of the row's loads are requested together and unconditionally (index clamped into the row; see the unquantised path)
        u32x4 kg[kChunks], ku[kChunks];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int j = min((c * 64 + lane) * N, half - N);
            if (c * 64 * N < half) {                          // (wave-uniform)
                kg[c] = *(const u32x4 *)(xr + j);
                ku[c] = *(const u32x4 *)(xr + half + j);
            }
        }
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            if (c * 64 * N < half) {                          // (wave-uniform; lanes past the row compute on the clamped piece and store nothing)
                float u[N];
                unpack16<DT>(kg[c], keep[c]);
                unpack16<DT>(ku[c], u);
                const bool live = (c * 64 + lane) * N < half;
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    keep[c][e] = act(keep[c][e], u[e]);
                    amax = fmaxf(amax, live ? fabsf(keep[c][e]) : 0.f);
                }
            }
        }
    } else if (vec) {
        for (int j = lane * N; j < half; j += 64 * N) {
            float g[N], u[N];
            load16<DT>(xr + j, g);
            load16<DT>(xr + half + j, u);
#pragma unroll
            for (int e = 0; e < N; ++e) amax = fmaxf(amax, fabsf(act(g[e], u[e])));
        }
    } else {
        for (int j = lane; j < half; j += 64) amax = fmaxf(amax, fabsf(value(j)));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    const float sc = ACT == 1 ? fmaxf(amax / 127.0f, 1e-30f) : amax / 127.0f;
    if (lane == 0) scale[row] = sc;
    rsc = 1.0f / sc;
    int8_t *o = (int8_t *)out + row * (long long)half;
    if (in_regs) {
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int j = (c * 64 + lane) * N;
            if (j < half) {
                uint32_t w[2] = {0u, 0u};
#pragma unroll
                for (int e = 0; e < N; ++e) w[e >> 2] |= (uint32_t)(quant(keep[c][e], sc) & 0xFF) << (8 * (e & 3));
                *(uint2 *)(o + j) = uint2{w[0], w[1]};
            }
        }
    } else if (vec) {                                      // second pass over the row (out of L2)
        for (int j = lane * N; j < half; j += 64 * N) {
            float g[N], u[N];
            load16<DT>(xr + j, g);
            load16<DT>(xr + half + j, u);
            uint32_t w[2] = {0u, 0u};
#pragma unroll
            for (int e = 0; e < N; ++e) w[e >> 2] |= (uint32_t)(quant(act(g[e], u[e]), sc) & 0xFF) << (8 * (e & 3));
            *(uint2 *)(o + j) = uint2{w[0], w[1]};
        }
    } else {
        for (int j = lane; j < half; j += 64) o[j] = (int8_t)quant(value(j), sc);
    }
}

// Kimi-K3 attention residual (kimi_k3/attn_residual.py:7-63): per token, B bank rows and the prefix row are scored -- score = sum(row *
// rsqrt(mean(row^2) + eps) * combined_weight) (:40-41) --, the scores go through a softmax (:43-45) and the output is the probability-
// weighted sum of the same rows (:47-59).  One wave per token, four tokens per workgroup; the rows are read a second time (out of L2) for the
// mix.  WT = type of combined_weight.
// sum over the 64 lanes on DPP (row_shr inside the rows of 16, row_bcast:15 / :31 across them; total in lane 63) instead of six
// ds_bpermute round trips: two reductions per bank row sit on this kernel's critical path
__device__ __forceinline__ float wave_sum_dpp(float v)
{
#define MI_DPP_ADD(CTRL, ROWS) v += __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROWS, 0xf, false))
    MI_DPP_ADD(0x111, 0xf);
    MI_DPP_ADD(0x112, 0xf);
    MI_DPP_ADD(0x114, 0xf);
    MI_DPP_ADD(0x118, 0xf);
    MI_DPP_ADD(0x142, 0xa);
    MI_DPP_ADD(0x143, 0xc);
#undef MI_DPP_ADD
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}

template <int DT, int WT>
__global__ __launch_bounds__(256) void attn_residual_mix_kernel(const typename Elem<DT>::T *__restrict__ prefix, long long stride_pm,
                                                                const typename Elem<DT>::T *__restrict__ bank, long long stride_bm, long long stride_bb,
                                                                const typename Elem<WT>::T *__restrict__ cw, long long tokens, int B, int H, float eps,
                                                                typename Elem<DT>::T *__restrict__ out, long long stride_om)
{
    typedef typename Elem<DT>::T T;
    constexpr int N = Elem<DT>::kPer16, kBatch = 4;          // 16-byte pieces requested together
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= tokens) return;
    auto row_ptr = [&](int r) -> const T * { return r < B ? bank + t * stride_bm + r * stride_bb : prefix + t * stride_pm; };
    const int nchunks = (H + 64 * N - 1) / (64 * N);
    // pass 1: per row, sum of squares and sum(row * weight) in one sweep (score = rsqrt(mean + eps) * sum(row * weight): the row-wide factor
    // taken out of the sum, so that no row has to be held); loads in batches of four pieces, index clamped, contributions masked
    float my_score = -INFINITY;                              // lane r keeps the score of row r
    for (int r = 0; r <= B; ++r) {
        const T *rp = row_ptr(r);
        float ss = 0.f, dot = 0.f;
        for (int c0 = 0; c0 < nchunks; c0 += kBatch) {
            u32x4 raw[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) raw[b] = *(const u32x4 *)(rp + min(((c0 + b) * 64 + lane) * N, H - N));
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const int j = ((c0 + b) * 64 + lane) * N;
                if (j < H) {
                    float v[N];
                    unpack16<DT>(raw[b], v);
#pragma unroll
                    for (int e = 0; e < N; ++e) {
                        ss += v[e] * v[e];
                        dot += v[e] * Elem<WT>::ld(cw[j + e]);
                    }
                }
            }
        }
        const float score = wave_sum_dpp(dot) * rsqrtf(wave_sum_dpp(ss) / (float)H + eps);
        if (lane == r) my_score = score;
    }
    float mx = my_score;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float ex = lane <= B ? __expf(my_score - mx) : 0.f;
    const float prob = ex / wave_sum_dpp(ex);
    // pass 2: piece by piece, the B + 1 rows of a piece mixed into eight accumulators (the rows come out of L2 this time)
    T *o = out + t * stride_om;
    for (int c = 0; c < nchunks; ++c) {
        const int j = (c * 64 + lane) * N, jc = min(j, H - N);
        float acc[N];
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = 0.f;
        for (int r0 = 0; r0 <= B; r0 += kBatch) {
            u32x4 raw[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) raw[b] = *(const u32x4 *)(row_ptr(min(r0 + b, B)) + jc);
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                if (r0 + b <= B) {                           // (wave-uniform)
                    const float p = __shfl(prob, r0 + b, 64);
                    float v[N];
                    unpack16<DT>(raw[b], v);
#pragma unroll
                    for (int e = 0; e < N; ++e) acc[e] += p * v[e];
                }
            }
        }
        if (j < H) store16<DT>(o + j, acc);
    }
}

// out = routed * factor + shared (moe/mul_add.py:9-36), the shared-expert add behind the MoE combine.  The product is rounded to the I/O
// dtype before the sum, as the tensor expression `routed * factor + shared` evaluates in that dtype (two roundings).
template <int DT>
__global__ __launch_bounds__(256) void mul_add_kernel(const typename Elem<DT>::T *__restrict__ a, const typename Elem<DT>::T *__restrict__ b, float factor,
                                                      long long numel, typename Elem<DT>::T *__restrict__ out)
{
    constexpr int N = Elem<DT>::kPer16;
    const long long stride = (long long)gridDim.x * 256 * N;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * N; i < numel; i += stride) {
        if (i + N <= numel) {
            float x[N], y[N];
            load16<DT>(a + i, x);
            load16<DT>(b + i, y);
#pragma unroll
            for (int e = 0; e < N; ++e) x[e] = Elem<DT>::ld(Elem<DT>::st(x[e] * factor)) + y[e];
            store16<DT>(out + i, x);
        } else {
            for (long long j = i; j < numel; ++j) out[j] = Elem<DT>::st(Elem<DT>::ld(Elem<DT>::st(Elem<DT>::ld(a[j]) * factor)) + Elem<DT>::ld(b[j]));
        }
    }
}

// "Zero experts" of type identity (moe/zero_experts_compute_identity.py:6-47): the selections of a token that point past the real experts
// (idx >= num_experts) contribute hidden * (sum of their scales) (:24-25, :35-39); their scales are then cleared and their indices replaced
// by identity_mask_value -- the first by 0 when ALL K selections were zero experts (:28-31, :40-41).  One wave per token; K <= 64.
template <int DT, bool I64, int ST>
__global__ __launch_bounds__(256) void zero_experts_identity_kernel(void *__restrict__ idx, typename Elem<ST>::T *__restrict__ scales, int num_experts,
                                                                    const typename Elem<DT>::T *__restrict__ hidden, long long tokens, int K, int D,
                                                                    int identity_mask_value, typename Elem<DT>::T *__restrict__ result)
{
    constexpr int N = Elem<DT>::kPer16;
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= tokens) return;
    long long e = 0;
    if (lane < K) e = I64 ? ((const long long *)idx)[t * K + lane] : (long long)((const int *)idx)[t * K + lane];
    const bool zero_expert = lane < K && e >= num_experts;
    float sc = zero_expert ? Elem<ST>::ld(scales[t * K + lane]) : 0.f;
    const float sum_scales = wave_sum_f(sc);
    const bool all_zero = __popcll(__ballot(zero_expert)) == K;
    const typename Elem<DT>::T *h = hidden + t * (long long)D;
    typename Elem<DT>::T *r = result + t * (long long)D;
    if (D % N == 0) {
        for (int j = lane * N; j < D; j += 64 * N) {
            float x[N];
            load16<DT>(h + j, x);
#pragma unroll
            for (int q = 0; q < N; ++q) x[q] = x[q] * sum_scales;
            store16<DT>(r + j, x);
        }
    } else {
        for (int j = lane; j < D; j += 64) r[j] = Elem<DT>::st(Elem<DT>::ld(h[j]) * sum_scales);
    }
    if (zero_expert) {
        scales[t * K + lane] = Elem<ST>::st(0.f);
        const long long v = (all_zero && lane == 0) ? 0 : (long long)identity_mask_value;
        if (I64) ((long long *)idx)[t * K + lane] = v;
        else ((int *)idx)[t * K + lane] = (int)v;
    }
}

// out = x * (c + scale) + shift (norm/scale_shift.py:9-183): scale one value or one per column, shift one value, one per column or one per
// element.  With a per-element shift c = scale_constant (fused_scale_shift_kernel_2, :112), otherwise c = 1.0 whatever scale_constant says
// (fused_scale_shift_kernel, :60) -- as the reference.  DT = type of x and out, ST = type of scale and shift; fp32 arithmetic.
template <int DT, int ST>
__global__ __launch_bounds__(256) void scale_shift_kernel(const typename Elem<DT>::T *__restrict__ x, const typename Elem<ST>::T *__restrict__ scale,
                                                          const typename Elem<ST>::T *__restrict__ shift, long long numel, int cols, int scale_numel,
                                                          long long shift_numel, float c, typename Elem<DT>::T *__restrict__ out)
{
    constexpr int N = Elem<DT>::kPer16;
    const bool vec = (cols % N) == 0;
    const long long stride = (long long)gridDim.x * 256;
    if (vec) {
        const long long chunks = numel / N;
        for (long long ch = (long long)blockIdx.x * 256 + threadIdx.x; ch < chunks; ch += stride) {
            const long long i0 = ch * N;
            const int col0 = (int)(i0 % cols);
            float f[N];
            load16<DT>(x + i0, f);
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float sc = Elem<ST>::ld(scale[scale_numel == 1 ? 0 : col0 + e]);
                const float sh = Elem<ST>::ld(shift[shift_numel == numel ? i0 + e : (shift_numel == 1 ? 0 : (long long)(col0 + e))]);
                f[e] = f[e] * (c + sc) + sh;
            }
            store16<DT>(out + i0, f);
        }
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < numel; i += stride) {
            const int col = (int)(i % cols);
            const float sc = Elem<ST>::ld(scale[scale_numel == 1 ? 0 : col]);
            const float sh = Elem<ST>::ld(shift[shift_numel == numel ? i : (shift_numel == 1 ? 0 : (long long)col)]);
            out[i] = Elem<DT>::st(Elem<DT>::ld(x[i]) * (c + sc) + sh);
        }
    }
}

// ---- split QKV + tensor-parallel RMSNorm + RoPE (norm/split_qkv_tp_rmsnorm_rope.py:7-288) -------------------------------------------------
// The norm runs over the WHOLE q row and the whole k row of this rank's shard (not per head), its mean of squares is summed over the
// tensor-parallel ranks between the two kernels (the caller's all-reduce).
//   kernel 1 (:7-73): V copied; qk_var[row] = {sum(q^2) / q_cols, sum(k^2) / k_cols} in fp32 (the reference also copies q and k here and
//            normalises them in place later; here kernel 2 reads them from the input row, same result, one pass less)
//   kernel 2 (:75-177): scale = 1 / sqrt(var * inv_tp_world + eps); y = dtype((x * scale) * w[col]) -- rounded to the I/O dtype before the
//            rotation, as the reference stores it (:113-117) --; per head, p < rotary_dim / 2: out[p] = y[p] cos[p] - y[p + h] sin[p],
//            out[p + h] = y[p + h] cos[p] + y[p] sin[p] with the FIRST half of the row's cos / sin (:131-152); other dims keep y.
template <int DT>
__global__ __launch_bounds__(256) void tp_var_kernel(const typename Elem<DT>::T *__restrict__ in, int q_cols, int k_cols, typename Elem<DT>::T *__restrict__ v,
                                                     float *__restrict__ qk_var)
{
    typedef typename Elem<DT>::T T;
    constexpr int N = Elem<DT>::kPer16;
    __shared__ float red[2][4];
    const long long row = blockIdx.x;
    const int tid = threadIdx.x;
    const T *rin = in + row * (long long)(q_cols + 2 * k_cols);
    float ssq = 0.f, ssk = 0.f;
    for (int i = tid * N; i < q_cols; i += 256 * N) {
        float f[N];
        load16<DT>(rin + i, f);
#pragma unroll
        for (int e = 0; e < N; ++e) ssq += f[e] * f[e];
    }
    for (int i = tid * N; i < k_cols; i += 256 * N) {
        float f[N];
        load16<DT>(rin + q_cols + i, f);
#pragma unroll
        for (int e = 0; e < N; ++e) ssk += f[e] * f[e];
        *(u32x4 *)(v + row * (long long)k_cols + i) = *(const u32x4 *)(rin + q_cols + k_cols + i);
    }
    ssq = wave_sum_f(ssq), ssk = wave_sum_f(ssk);
    if ((tid & 63) == 0) red[0][tid >> 6] = ssq, red[1][tid >> 6] = ssk;
    __syncthreads();
    if (tid == 0) {
        qk_var[row * 2 + 0] = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) / (float)q_cols;
        qk_var[row * 2 + 1] = (((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / (float)k_cols;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void tp_norm_rope_kernel(const typename Elem<DT>::T *__restrict__ in, const typename Elem<DT>::T *__restrict__ cos,
                                                           const typename Elem<DT>::T *__restrict__ sin, const float *__restrict__ qk_var, int q_cols, int k_cols,
                                                           int head_dim, int rotary_dim, float eps, float inv_tp, const typename Elem<DT>::T *__restrict__ qw,
                                                           const typename Elem<DT>::T *__restrict__ kw, typename Elem<DT>::T *__restrict__ q,
                                                           typename Elem<DT>::T *__restrict__ k)
{
    typedef typename Elem<DT>::T T;
    constexpr int N = Elem<DT>::kPer16;
    const long long row = blockIdx.x;
    const int tid = threadIdx.x;
    const T *rin = in + row * (long long)(q_cols + 2 * k_cols);
    const int half = rotary_dim >> 1;
    const T *cr = cos + row * (long long)rotary_dim, *sr = sin + row * (long long)rotary_dim;
    // units of N consecutive elements: a rotated unit carries its partner (half further on) with it
    const int units_head = head_dim / N, rot_units = half / N;               // per head: [0, rot_units) lower halves, partners implied
    const int work_head = units_head - rot_units;                           // lower-half units + the units behind rotary_dim
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        const int cols = part ? k_cols : q_cols;
        const T *src = part ? rin + q_cols : rin;
        const T *w = part ? kw : qw;
        T *dst = (part ? k + row * (long long)k_cols : q + row * (long long)q_cols);
        const float scale = 1.0f / sqrtf(qk_var[row * 2 + part] * inv_tp + eps);
        const int heads = cols / head_dim;
        for (int u = tid; u < heads * work_head; u += 256) {
            const int h = u / work_head, j = u - h * work_head;
            const bool rot = j < rot_units;
            const int c0 = h * head_dim + (rot ? j * N : rotary_dim + (j - rot_units) * N);
            float a[N], wa[N];
            load16<DT>(src + c0, a);
            load16<DT>(w + c0, wa);
#pragma unroll
            for (int e = 0; e < N; ++e) a[e] = Elem<DT>::ld(Elem<DT>::st((a[e] * scale) * wa[e]));
            if (rot) {
                float b[N], wb[N], cv[N], sv[N];
                load16<DT>(src + c0 + half, b);
                load16<DT>(w + c0 + half, wb);
                load16<DT>(cr + j * N, cv);
                load16<DT>(sr + j * N, sv);
                float o1[N], o2[N];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    b[e] = Elem<DT>::ld(Elem<DT>::st((b[e] * scale) * wb[e]));
                    o1[e] = a[e] * cv[e] - b[e] * sv[e];
                    o2[e] = b[e] * cv[e] + a[e] * sv[e];
                }
                store16<DT>(dst + c0, o1);
                store16<DT>(dst + c0 + half, o2);
            } else {
                store16<DT>(dst + c0, a);
            }
        }
    }
}

}  // namespace
}  // namespace mi_sgl

using namespace mi_sgl;

extern "C" int mi_fused_split_qk_norm(const void *x, long long rows, int q_lora_rank, int kv_lora_rank, int qk_rope_dim, float eps, const void *q_weight,
                                      const void *q_bias, const void *k_weight, const void *k_bias, int dtype, void *q_lora, void *k_nope, void *k_pe,
                                      void *stream)
{
    if (rows < 0 || rows >= (1ll << 31) || q_lora_rank <= 0 || kv_lora_rank <= 0 || qk_rope_dim <= 0 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16 && dtype != MI_DTYPE_F32))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!x || !q_weight || !k_weight || !q_lora || !k_nope || !k_pe) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define MI_SQK(DT)                                                                                                                           \
    split_qk_norm_kernel<DT><<<(unsigned)rows, 128, 0, st>>>((const typename Elem<DT>::T *)x, q_lora_rank, kv_lora_rank, qk_rope_dim, eps,      \
                                                             (const typename Elem<DT>::T *)q_weight, (const typename Elem<DT>::T *)q_bias,     \
                                                             (const typename Elem<DT>::T *)k_weight, (const typename Elem<DT>::T *)k_bias,     \
                                                             (typename Elem<DT>::T *)q_lora, (typename Elem<DT>::T *)k_nope, (typename Elem<DT>::T *)k_pe)
    if (dtype == MI_DTYPE_BF16) MI_SQK(MI_DTYPE_BF16);
    else if (dtype == MI_DTYPE_F16) MI_SQK(MI_DTYPE_F16);
    else MI_SQK(MI_DTYPE_F32);
#undef MI_SQK
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_swiglu_oai(const void *x, long long rows, int dim, float alpha, float limit, int dtype, void *out, void *stream)
{
    if (rows < 0 || dim <= 0 || dim % 2 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16 && dtype != MI_DTYPE_F32)) return MI_SGL_EINVAL;
    const int n16 = dtype == MI_DTYPE_F32 ? 4 : 8;
    if ((dim / 2) % n16) return MI_SGL_EINVAL;              // 16-byte pieces of the output rows
    const long long n_out = rows * (long long)(dim / 2);
    if (n_out == 0) return MI_SGL_OK;
    if (!x || !out) return MI_SGL_EINVAL;
    long long blocks = (n_out / n16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI_DTYPE_BF16) swiglu_oai_kernel<MI_DTYPE_BF16><<<(unsigned)blocks, 256, 0, st>>>((const uint16_t *)x, n_out, alpha, limit, (uint16_t *)out);
    else if (dtype == MI_DTYPE_F16) swiglu_oai_kernel<MI_DTYPE_F16><<<(unsigned)blocks, 256, 0, st>>>((const uint16_t *)x, n_out, alpha, limit, (uint16_t *)out);
    else swiglu_oai_kernel<MI_DTYPE_F32><<<(unsigned)blocks, 256, 0, st>>>((const float *)x, n_out, alpha, limit, (float *)out);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_swiglu_oai_quant(const void *x, const void *group_list, int group_list_is_i64, int num_groups, int group_list_type, long long rows,
                                   int cols, float alpha, float limit, int need_quant, int dtype, void *out, float *scale, void *stream)
{
    if (rows < 0 || cols <= 0 || cols % 2 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) || (group_list && (num_groups <= 0 || (group_list_type != 0 && group_list_type != 1))))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!x || !out || (need_quant && !scale) || rows > (1ll << 32)) return MI_SGL_EINVAL;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define MI_SOQ(DT, I64) swiglu_oai_quant_kernel<DT, I64><<<blocks, 256, 0, st>>>((const uint16_t *)x, group_list, num_groups, group_list_type, rows, cols / 2, \
                                                                             alpha, limit, need_quant, out, scale)
    if (dtype == MI_DTYPE_BF16) { if (group_list_is_i64) MI_SOQ(MI_DTYPE_BF16, true); else MI_SOQ(MI_DTYPE_BF16, false); }
    else { if (group_list_is_i64) MI_SOQ(MI_DTYPE_F16, true); else MI_SOQ(MI_DTYPE_F16, false); }
#undef MI_SOQ
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

// SiTU + optional per-row INT8 (activation/situ.py): x [rows, cols] = [gate | up]; linear_beta <= 0: the up path is left alone
extern "C" int mi_situ_and_mul(const void *x, const void *group_list, int group_list_is_i64, int num_groups, int group_list_type, long long rows,
                               int cols, float beta, float linear_beta, int need_quant, int dtype, void *out, float *scale, void *stream)
{
    if (rows < 0 || cols <= 0 || cols % 2 || !(beta > 0.f) || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) ||
        (group_list && (num_groups <= 0 || (group_list_type != 0 && group_list_type != 1))))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!x || !out || (need_quant && !scale) || rows > (1ll << 32)) return MI_SGL_EINVAL;
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define MI_SITU(DT, I64) swiglu_oai_quant_kernel<DT, I64, 1><<<blocks, 256, 0, st>>>((const uint16_t *)x, group_list, num_groups, group_list_type, rows, cols / 2, \
                                                                                 beta, linear_beta, need_quant, out, scale)
    if (dtype == MI_DTYPE_BF16) { if (group_list_is_i64) MI_SITU(MI_DTYPE_BF16, true); else MI_SITU(MI_DTYPE_BF16, false); }
    else { if (group_list_is_i64) MI_SITU(MI_DTYPE_F16, true); else MI_SITU(MI_DTYPE_F16, false); }
#undef MI_SITU
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_attn_residual_mix(const void *prefix_sum, long long stride_pm, const void *bank, long long stride_bm, long long stride_bb,
                                    const void *combined_weight, int weight_dtype, long long tokens, int num_valid_blocks, int hidden, float eps,
                                    int dtype, void *out, long long stride_om, void *stream)
{
    if (tokens < 0 || tokens >= (1ll << 31) || num_valid_blocks < 0 || num_valid_blocks > 63 || hidden <= 0 || hidden % 8 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) || (weight_dtype != dtype && weight_dtype != MI_DTYPE_F32))
        return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!prefix_sum || (!bank && num_valid_blocks > 0) || !combined_weight || !out) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define MI_AR(DT, WT)                                                                                                                         \
    attn_residual_mix_kernel<DT, WT><<<(unsigned)((tokens + 3) / 4), 256, 0, st>>>((const uint16_t *)prefix_sum, stride_pm, (const uint16_t *)bank,  \
                                                                                   stride_bm, stride_bb, (const typename Elem<WT>::T *)combined_weight, \
                                                                                   tokens, num_valid_blocks, hidden, eps, (uint16_t *)out, stride_om)
    if (dtype == MI_DTYPE_BF16) { if (weight_dtype == MI_DTYPE_F32) MI_AR(MI_DTYPE_BF16, MI_DTYPE_F32); else MI_AR(MI_DTYPE_BF16, MI_DTYPE_BF16); }
    else { if (weight_dtype == MI_DTYPE_F32) MI_AR(MI_DTYPE_F16, MI_DTYPE_F32); else MI_AR(MI_DTYPE_F16, MI_DTYPE_F16); }
#undef MI_AR
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_mul_add(const void *routed, const void *shared, float factor, long long numel, int dtype, void *out, void *stream)
{
    if (numel < 0 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16 && dtype != MI_DTYPE_F32)) return MI_SGL_EINVAL;
    if (numel == 0) return MI_SGL_OK;
    if (!routed || !shared || !out) return MI_SGL_EINVAL;
    long long blocks = (numel / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 256 * 32 ? 256 * 32 : blocks);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI_DTYPE_BF16) mul_add_kernel<MI_DTYPE_BF16><<<(unsigned)blocks, 256, 0, st>>>((const uint16_t *)routed, (const uint16_t *)shared, factor, numel, (uint16_t *)out);
    else if (dtype == MI_DTYPE_F16) mul_add_kernel<MI_DTYPE_F16><<<(unsigned)blocks, 256, 0, st>>>((const uint16_t *)routed, (const uint16_t *)shared, factor, numel, (uint16_t *)out);
    else mul_add_kernel<MI_DTYPE_F32><<<(unsigned)blocks, 256, 0, st>>>((const float *)routed, (const float *)shared, factor, numel, (float *)out);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_zero_experts_identity(void *expert_indices, int idx_is_i64, void *expert_scales, int scales_dtype, int num_experts,
                                        const void *hidden, long long tokens, int K, int D, int identity_mask_value, int dtype, void *result,
                                        void *stream)
{
    if (tokens < 0 || K <= 0 || K > 64 || D <= 0 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) ||
        (scales_dtype != MI_DTYPE_F32 && scales_dtype != dtype) || tokens > (1ll << 32))
        return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!expert_indices || !expert_scales || !hidden || !result) return MI_SGL_EINVAL;
    const unsigned blocks = (unsigned)((tokens + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
#define MI_ZE(DT, I64, ST)                                                                                                                            \
    zero_experts_identity_kernel<DT, I64, ST><<<blocks, 256, 0, st>>>(expert_indices, (typename Elem<ST>::T *)expert_scales, num_experts,             \
                                                                      (const uint16_t *)hidden, tokens, K, D, identity_mask_value, (uint16_t *)result)
#define MI_ZE2(DT)                                                                                                                \
    do {                                                                                                                          \
        if (scales_dtype == MI_DTYPE_F32) { if (idx_is_i64) MI_ZE(DT, true, MI_DTYPE_F32); else MI_ZE(DT, false, MI_DTYPE_F32); } \
        else { if (idx_is_i64) MI_ZE(DT, true, DT); else MI_ZE(DT, false, DT); }                                                  \
    } while (0)
    if (dtype == MI_DTYPE_BF16) MI_ZE2(MI_DTYPE_BF16); else MI_ZE2(MI_DTYPE_F16);
#undef MI_ZE2
#undef MI_ZE
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

static bool tp_shape_ok(long long rows, int q_cols, int k_cols, int head_dim, int rotary_dim, int dtype)
{
    const int n = dtype == MI_DTYPE_F32 ? 4 : 8;
    return rows >= 0 && rows < (1ll << 31) && head_dim > 0 && (head_dim & (head_dim - 1)) == 0 && q_cols > 0 && k_cols > 0 && q_cols % head_dim == 0 &&
           k_cols % head_dim == 0 && q_cols % k_cols == 0 && rotary_dim > 0 && rotary_dim <= head_dim && rotary_dim % (2 * n) == 0 && head_dim % n == 0 &&
           (dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16 || dtype == MI_DTYPE_F32);
}

extern "C" int mi_split_qkv_tp_var(const void *input, long long rows, int q_cols, int k_cols, int dtype, void *v, float *qk_var, void *stream)
{
    const int n16 = dtype == MI_DTYPE_F32 ? 4 : 8;          // elements per 16-byte access
    if (rows < 0 || rows >= (1ll << 31) || q_cols <= 0 || k_cols <= 0 || q_cols % n16 || k_cols % n16 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16 && dtype != MI_DTYPE_F32))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!input || !v || !qk_var) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI_DTYPE_BF16) tp_var_kernel<MI_DTYPE_BF16><<<(unsigned)rows, 256, 0, st>>>((const uint16_t *)input, q_cols, k_cols, (uint16_t *)v, qk_var);
    else if (dtype == MI_DTYPE_F16) tp_var_kernel<MI_DTYPE_F16><<<(unsigned)rows, 256, 0, st>>>((const uint16_t *)input, q_cols, k_cols, (uint16_t *)v, qk_var);
    else tp_var_kernel<MI_DTYPE_F32><<<(unsigned)rows, 256, 0, st>>>((const float *)input, q_cols, k_cols, (float *)v, qk_var);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_split_qkv_tp_norm_rope(const void *input, const void *cos, const void *sin, const float *qk_var, long long rows, int q_cols, int k_cols,
                                         int head_dim, int rotary_dim, float eps, float inv_tp_world, const void *q_weight, const void *k_weight, int dtype,
                                         void *q, void *k, void *stream)
{
    if (!tp_shape_ok(rows, q_cols, k_cols, head_dim, rotary_dim, dtype)) return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!input || !cos || !sin || !qk_var || !q_weight || !k_weight || !q || !k) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define MI_TPN(DT)                                                                                                                              \
    tp_norm_rope_kernel<DT><<<(unsigned)rows, 256, 0, st>>>((const typename Elem<DT>::T *)input, (const typename Elem<DT>::T *)cos,               \
                                                            (const typename Elem<DT>::T *)sin, qk_var, q_cols, k_cols, head_dim, rotary_dim, eps,  \
                                                            inv_tp_world, (const typename Elem<DT>::T *)q_weight, (const typename Elem<DT>::T *)k_weight, \
                                                            (typename Elem<DT>::T *)q, (typename Elem<DT>::T *)k)
    if (dtype == MI_DTYPE_BF16) MI_TPN(MI_DTYPE_BF16);
    else if (dtype == MI_DTYPE_F16) MI_TPN(MI_DTYPE_F16);
    else MI_TPN(MI_DTYPE_F32);
#undef MI_TPN
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_scale_shift(const void *x, const void *scale, const void *shift, long long rows, int cols, long long scale_numel,
                              long long shift_numel, float scale_constant, int dtype, int ss_dtype, void *out, void *stream)
{
    auto ok_dt = [](int d) { return d == MI_DTYPE_BF16 || d == MI_DTYPE_F16 || d == MI_DTYPE_F32; };
    if (rows < 0 || cols <= 0 || !ok_dt(dtype) || (ss_dtype != dtype && ss_dtype != MI_DTYPE_F32)) return MI_SGL_EINVAL;
    const long long numel = rows * cols;
    // the reference's asserts (scale_shift.py:136-141); a per-element shift goes with a per-column scale (kernel_2 reads scale by column)
    if ((scale_numel != 1 && scale_numel != cols) || (shift_numel != 1 && shift_numel != cols && shift_numel != numel)) return MI_SGL_EINVAL;
    const bool full_shift = shift_numel == numel;           // tested first, as the reference does (:149): a one-row x takes this form too
    if (full_shift && scale_numel != cols) return MI_SGL_EINVAL;
    if (numel == 0) return MI_SGL_OK;
    if (!x || !scale || !shift || !out) return MI_SGL_EINVAL;
    const float c = full_shift ? scale_constant : 1.0f;
    long long blocks = (numel / 4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
#define MI_SS(DT, ST)                                                                                                                     \
    scale_shift_kernel<DT, ST><<<(unsigned)blocks, 256, 0, st>>>((const typename Elem<DT>::T *)x, (const typename Elem<ST>::T *)scale,    \
                                                                 (const typename Elem<ST>::T *)shift, numel, cols, (int)scale_numel,      \
                                                                 shift_numel, c, (typename Elem<DT>::T *)out)
    if (dtype == MI_DTYPE_F32) MI_SS(MI_DTYPE_F32, MI_DTYPE_F32);
    else if (dtype == MI_DTYPE_BF16) { if (ss_dtype == MI_DTYPE_F32) MI_SS(MI_DTYPE_BF16, MI_DTYPE_F32); else MI_SS(MI_DTYPE_BF16, MI_DTYPE_BF16); }
    else { if (ss_dtype == MI_DTYPE_F32) MI_SS(MI_DTYPE_F16, MI_DTYPE_F32); else MI_SS(MI_DTYPE_F16, MI_DTYPE_F16); }
#undef MI_SS
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_l1_norm(const void *x, long long rows, int cols, int dtype, float *out, void *stream)
{
    if (cols > 0 && cols <= 32 && rows > 0 && rows <= (1ll << 32) && x && out && (dtype == MI_DTYPE_BF16 || dtype == MI_DTYPE_F16 || dtype == MI_DTYPE_F32)) {
        const unsigned blocks = (unsigned)((rows + 255) / 256);
        hipStream_t st = (hipStream_t)stream;
        if (dtype == MI_DTYPE_BF16) l1_small_kernel<MI_DTYPE_BF16><<<blocks, 256, 0, st>>>((const uint16_t *)x, rows, cols, out);
        else if (dtype == MI_DTYPE_F16) l1_small_kernel<MI_DTYPE_F16><<<blocks, 256, 0, st>>>((const uint16_t *)x, rows, cols, out);
        else l1_small_kernel<MI_DTYPE_F32><<<blocks, 256, 0, st>>>((const float *)x, rows, cols, out);
        return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
    }
    return launch_rows<kOpL1>(x, rows, cols, 0.f, nullptr, nullptr, dtype, out, stream);
}

extern "C" int mi_rmsnorm_without_weight(const void *x, long long rows, int cols, float eps, int dtype, void *out, void *stream)
{
    return launch_rows<kOpRms>(x, rows, cols, eps, nullptr, nullptr, dtype, out, stream);
}

extern "C" int mi_row_variance(const void *x, long long rows, int cols, int dtype, void *out, void *stream)
{
    return launch_rows<kOpVariance>(x, rows, cols, 0.f, nullptr, nullptr, dtype, out, stream);
}

extern "C" int mi_rsqrt_mul(const void *x, const void *variance, const void *weight, long long rows, int cols, float eps, int dtype, void *out,
                            void *stream)
{
    return launch_rows<kOpRsqrtMul>(x, rows, cols, eps, variance, weight, dtype, out, stream);
}
