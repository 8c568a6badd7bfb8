// HBM-bound fused elementwise primitives for gfx950: SwiGLU + per-row INT8 quantisation, Add + RMSNorm (+bias, +static
// INT8 quantisation, Gemma variant), split-QKV + per-head RMSNorm + RoPE.
// Replace the reference's Triton-Ascend kernels:
//   swiglu_quant            python/sgl_kernel_npu/sgl_kernel_npu/activation/swiglu_quant.py:8-127
//   add_rmsnorm_bias        python/sgl_kernel_npu/sgl_kernel_npu/norm/add_rmsnorm_bias.py:8-147
//   add_gemma_rms_norm      python/sgl_kernel_npu/sgl_kernel_npu/norm/add_rmsnorm_bias.py:150-232
//   split_qkv_rmsnorm_rope  python/sgl_kernel_npu/sgl_kernel_npu/norm/split_qkv_rmsnorm_rope.py:8-438
// MI355X design: every row is read once with 16-B loads and held in registers in fp32 (no second pass over HBM for the
// reduction), one wave64 per row / head where the row fits 64 registers, one 256-thread workgroup per row for hidden
// sizes up to 8192; reductions are wave shuffles (+ one LDS hop across waves).  Algorithmic bytes are listed per op.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi_sgl_kernels.h"

namespace mi_sgl {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <bool BF16>
__device__ __forceinline__ float ld16(uint32_t bits)
{
    if constexpr (BF16) return __uint_as_float(bits << 16);
    else return (float)__builtin_bit_cast(_Float16, (uint16_t)bits);
}
template <bool BF16>
__device__ __forceinline__ uint32_t st16(float f)
{
    if constexpr (BF16) {
        return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)f);      // v_cvt_pk_bf16_f32: round to nearest even in one instruction
    } else {
        return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f);
    }
}
template <bool BF16>
__device__ __forceinline__ void unpack8(const u32x4 &v, float *f)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = ld16<BF16>(v[j] & 0xFFFFu);
        f[2 * j + 1] = ld16<BF16>(v[j] >> 16);
    }
}
template <bool BF16>
__device__ __forceinline__ u32x4 pack8(const float *f)
{
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (BF16) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            v[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{f[2 * j], f[2 * j + 1]}, bf16x2));   // one v_cvt_pk_bf16_f32
        } else {
            v[j] = st16<BF16>(f[2 * j]) | (st16<BF16>(f[2 * j + 1]) << 16);
        }
    }
    return v;
}
// v of the lane selected by DPP control CTRL (quad_perm / row_ror / row_mirror ...), all rows and banks enabled
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
    // maximum over the wave without LDS: four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
    // row_mirror), then v_permlane16_swap / v_permlane32_swap across the rows.  The shuffle form was six ds_bpermute round trips
    // (~120 cycles each), on the critical path of every token wave of the stage kernels.
#define MI_DPP_MAX(CTRL) v = fmaxf(v, __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false)))
    MI_DPP_MAX(0xB1);
    MI_DPP_MAX(0x4E);
    MI_DPP_MAX(0x141);
    MI_DPP_MAX(0x140);
#undef MI_DPP_MAX
    typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
    u32x2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ int sat_i8(float v)
{
    const float r = rintf(v);
    return (int)fminf(fmaxf(r, -128.f), 127.f);
}

// ------------------------------------------------------------------------------------------------
// SwiGLU + per-row symmetric INT8:  out = x1 * sigmoid(x1) * x2 (fp32), scale = max|out| / 127,
// q = clamp(floor(out / scale + 0.5), -128, 127)      (swiglu_quant.py:49-72; note floor(x+0.5), not rint)
// bytes per row: 2I*2 read + I (+4) written.  One wave per row, I <= 4096.
// ------------------------------------------------------------------------------------------------
constexpr int kSwigluItems = 8;   // at most 8 x (64 lanes x 8 elements) = 4096 output columns per row

// ITEMS = 16-byte items per lane (1, 2, 4 or 8): the row's SwiGLU values stay in registers between the max and the
// quantisation pass, so the instantiation is picked by row width to keep the footprint (and occupancy) tight.
template <bool BF16, bool I64, int ITEMS>
__global__ __launch_bounds__(256) void swiglu_quant_kernel(const uint16_t *__restrict__ x, const void *__restrict__ group_list,
                                                          int num_groups, int group_list_type, int rows, int I,
                                                          int need_quant, int do_limit, float limit, void *__restrict__ out,
                                                          float *__restrict__ scale)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    // number of valid rows: last cumulative entry (type 0) or the sum of the per-group counts (type 1)
    long long total;
    if (group_list_type == 0) {
        // the reference indexes one past the end here (swiglu_quant.py:27); the cumulative total is the last entry
        total = I64 ? ((const long long *)group_list)[num_groups - 1] : (long long)((const int *)group_list)[num_groups - 1];
    } else {
        long long s = 0;
        for (int i = lane; i < num_groups; i += 64) s += I64 ? ((const long long *)group_list)[i] : (long long)((const int *)group_list)[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        total = s;
    }
    if (row >= rows || row >= total) return;
    const u32x4 *xr = (const u32x4 *)(x + row * 2 * (long long)I);
    const int nitems = I / 8;
    float v[ITEMS][8];
    float amax = 0.f;
    // all loads of the row first, unconditional (index clamped into the row): under `if (item < nitems)` every item's pair of loads sat
    // in its own block behind an s_waitcnt vmcnt(0), i.e. up to eight serial memory round trips per row and wave
    u32x4 ra[ITEMS], rb[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = min(it * 64 + lane, nitems - 1);
        ra[it] = xr[item];
        rb[it] = xr[nitems + item];
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = it * 64 + lane;
        if (item < nitems) {
            float a[8], b[8];
            unpack8<BF16>(ra[it], a);
            unpack8<BF16>(rb[it], b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // v_exp + v_rcp (1 ulp each) instead of an IEEE division: the reference test allows |dq| <= 1 on < 2 % of the
                // elements (test_swiglu_quant.py:45-54), and the two divisions per element made this kernel VALU-bound
                float gate = a[j] * __builtin_amdgcn_rcpf(1.0f + __expf(-a[j]));
                float up = b[j];
                if (do_limit) {
                    gate = fminf(gate, limit);
                    up = fmaxf(fminf(up, limit), -limit);
                }
                v[it][j] = gate * up;
                amax = fmaxf(amax, fabsf(v[it][j]));
            }
        }
    }
    if (!need_quant) {
        u32x4 *orow = (u32x4 *)((uint16_t *)out + row * (long long)I);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int item = it * 64 + lane;
            if (item < nitems) orow[item] = pack8<BF16>(v[it]);
        }
        return;
    }
    amax = wave_max(amax);
    const float s = amax / 127.0f;
    const float inv_s = s > 0.f ? 127.0f / amax : 0.f;
    if (lane == 0) scale[row] = s;
    u32x2 *orow = (u32x2 *)((int8_t *)out + row * (long long)I);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int item = it * 64 + lane;
        if (item < nitems) {
            uint32_t w[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float q = floorf(v[it][j] * inv_s + 0.5f);
                q = fminf(fmaxf(q, -128.f), 127.f);
                w[j >> 2] |= ((uint32_t)((int)q & 0xFF)) << (8 * (j & 3));
            }
            orow[item] = u32x2{w[0], w[1]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Add + RMSNorm (+ bias) (+ static INT8 quantisation) / Gemma variant.  One 256-thread workgroup per row, H <= 8192.
//   y = in (+ res) rounded to the I/O dtype (stored as out2);  v = float(y) * rstd * w (+ b)   [gemma: * (w + 1)]
//   out = v in the I/O dtype, or int8_sat(rint(v * qscale + qoffset))      (add_rmsnorm_bias.py:33-68,185-190)
// bytes per row: 2H in (+2H res) + 2H out2 + 2H (or H) out.
// ------------------------------------------------------------------------------------------------
constexpr int kNormItems = 4;     // 4 x (256 threads x 8 elements) = 8192 columns

template <bool BF16>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const uint16_t *__restrict__ in, const uint16_t *__restrict__ res,
                                                         const uint16_t *__restrict__ w, const uint16_t *__restrict__ bias,
                                                         const uint16_t *__restrict__ qs, const uint16_t *__restrict__ qo,
                                                         float eps, int gemma, int H, long long in_stride,
                                                         void *__restrict__ out, uint16_t *__restrict__ out2)
{
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const int tid = threadIdx.x;
    const int nitems = H / 8;
    const u32x4 *ir = (const u32x4 *)(in + row * in_stride);
    const u32x4 *rr = res ? (const u32x4 *)(res + row * in_stride) : nullptr;
    float y[kNormItems][8];
    float ss = 0.f;
    // every load of the row first, unconditional (index clamped into the row), the weights with them: under `if (item < nitems)` each
    // item's loads sat in their own block behind an s_waitcnt vmcnt(0), and the weight loads waited behind the reduction barrier
    u32x4 ra[kNormItems], rb[kNormItems], rw[kNormItems];
#pragma unroll
    for (int it = 0; it < kNormItems; ++it) {
        const int item = min(it * 256 + tid, nitems - 1);
        ra[it] = ir[item];
        if (rr) rb[it] = rr[item];
        rw[it] = ((const u32x4 *)w)[item];
    }
#pragma unroll
    for (int it = 0; it < kNormItems; ++it) {
        const int item = it * 256 + tid;
        if (item < nitems) {
            float a[8];
            unpack8<BF16>(ra[it], a);
            if (rr) {
                float b[8];
                unpack8<BF16>(rb[it], b);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = ld16<BF16>(st16<BF16>(a[j] + b[j]));   // the sum lives in the I/O dtype
                if (out2) ((u32x4 *)(out2 + row * (long long)H))[item] = pack8<BF16>(a);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                y[it][j] = a[j];
                ss += a[j] * a[j];
            }
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float var = (red[0] + red[1] + red[2] + red[3]) / (float)H;
    const float rstd = gemma ? rsqrtf(var + eps) : 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int it = 0; it < kNormItems; ++it) {
        const int item = it * 256 + tid;
        if (item < nitems) {
            float wv[8], o[8];
            unpack8<BF16>(rw[it], wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (y[it][j] * rstd) * (gemma ? wv[j] + 1.0f : wv[j]);
            if (bias) {
                float bv[8];
                unpack8<BF16>(((const u32x4 *)bias)[item], bv);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = o[j] + bv[j];
            }
            if (qs) {
                float sv[8], ov[8];
                unpack8<BF16>(((const u32x4 *)qs)[item], sv);
                unpack8<BF16>(((const u32x4 *)qo)[item], ov);
                uint32_t wq[2] = {0, 0};
#pragma unroll
                for (int j = 0; j < 8; ++j) wq[j >> 2] |= ((uint32_t)(sat_i8(o[j] * sv[j] + ov[j]) & 0xFF)) << (8 * (j & 3));
                ((u32x2 *)((int8_t *)out + row * (long long)H))[item] = u32x2{wq[0], wq[1]};
            } else {
                ((u32x4 *)((uint16_t *)out + row * (long long)H))[item] = pack8<BF16>(o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// split QKV + per-head RMSNorm (+bias) + RoPE on the first rope_dim dims; V is copied.  One wave per (row, head).
//   neox:        out[p] = x[p]*cos[p] - x[p+h]*sin[p],  out[p+h] = x[p+h]*cos[p+h] + x[p]*sin[p+h]          (h = rope_dim/2)
//   interleaved: out[2i] = x[2i]*cos[i] - x[2i+1]*sin[i], out[2i+1] = x[2i+1]*cos[i] + x[2i]*sin[i]
// (split_qkv_rmsnorm_rope.py:38-198; sin / cos rows are [rope_dim] per batch row)
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void split_qkv_rmsnorm_rope_kernel(
    const uint16_t *__restrict__ qkv, const uint16_t *__restrict__ sin, const uint16_t *__restrict__ cos, int rows, int q_hidden,
    int kv_hidden, int head_dim, int rope_dim, int has_norm, float eps, const uint16_t *__restrict__ qw,
    const uint16_t *__restrict__ kw, const uint16_t *__restrict__ qb, const uint16_t *__restrict__ kb, int neox,
    uint16_t *__restrict__ q, uint16_t *__restrict__ k, uint16_t *__restrict__ v)
{
    const int lane = threadIdx.x & 63;
    const int q_heads = q_hidden / head_dim, kv_heads = kv_hidden / head_dim;
    const int heads_total = q_heads + 2 * kv_heads;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long row = wid / heads_total;
    const int h = (int)(wid % heads_total);
    if (row >= rows) return;
    const long long total_hidden = (long long)q_hidden + 2ll * kv_hidden;
    const uint16_t *src = qkv + row * total_hidden + (long long)h * head_dim;
    if (h >= q_heads + kv_heads) {      // V: plain copy
        uint16_t *dst = v + row * (long long)kv_hidden + (long long)(h - q_heads - kv_heads) * head_dim;
        for (int i = lane; i < head_dim; i += 64) dst[i] = src[i];
        return;
    }
    const bool is_q = h < q_heads;
    uint16_t *dst = is_q ? q + row * (long long)q_hidden + (long long)h * head_dim
                         : k + row * (long long)kv_hidden + (long long)(h - q_heads) * head_dim;
    const uint16_t *wt = is_q ? qw : kw, *bs = is_q ? qb : kb;
    float ss = 0.f;
    if (has_norm)
        for (int i = lane; i < head_dim; i += 64) {
            const float a = ld16<BF16>(src[i]);
            ss += a * a;
        }
    float rstd = 1.f;
    if (has_norm) rstd = 1.0f / sqrtf(wave_sum(ss) / (float)head_dim + eps);
    auto nrm = [&](int i) -> float {
        float a = ld16<BF16>(src[i]);
        if (has_norm) {
            a = (a * rstd) * ld16<BF16>(wt[i]);
            if (bs) a = a + ld16<BF16>(bs[i]);
        }
        return a;
    };
    const int half = rope_dim / 2;
    const uint16_t *sr = sin + row * (long long)rope_dim, *cr = cos + row * (long long)rope_dim;
    for (int p = lane; p < half; p += 64) {
        if (neox) {
            const float x1 = nrm(p), x2 = nrm(p + half);
            dst[p] = (uint16_t)st16<BF16>((-x2) * ld16<BF16>(sr[p]) + x1 * ld16<BF16>(cr[p]));
            dst[p + half] = (uint16_t)st16<BF16>(x1 * ld16<BF16>(sr[p + half]) + x2 * ld16<BF16>(cr[p + half]));
        } else {
            const float x1 = nrm(2 * p), x2 = nrm(2 * p + 1);
            const float s = ld16<BF16>(sr[p]), c = ld16<BF16>(cr[p]);
            dst[2 * p] = (uint16_t)st16<BF16>((-x2) * s + x1 * c);
            dst[2 * p + 1] = (uint16_t)st16<BF16>(x1 * s + x2 * c);
        }
    }
    for (int i = rope_dim + lane; i < head_dim; i += 64) dst[i] = (uint16_t)st16<BF16>(nrm(i));
}

// ------------------------------------------------------------------------------------------------
// split [q | gate] pairs + K + V, Gemma RMSNorm (weight + 1) of every q and k head, neox RoPE on the first rope_dim dims
// (split_qkv_rmsnorm_rope.py:441-745, split_qkvgate_gemma_rmsnorm_rope).  Input row = q_heads x [q head | gate head], then K, then V
// (:478-497, :590-592, :660).  One wave per (row, item), items = q heads (q + its gate), k heads, v heads; a lane owns 8 consecutive
// elements (16-byte loads / stores) of up to four chunks of the head.
//   q, k:  y = (x * rsqrt(mean(x^2) + eps)) * (w + 1)   in fp32 (:499-503, :603-609)
//          out[p] = (-y[p + h]) * sin[p] + y[p] * cos[p],  out[p + h] = y[p] * sin[p + h] + y[p + h] * cos[p + h],  h = rope_dim / 2 (:505-558)
//   gate, v: copied (:567-571, :658-667)
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void split_qkvgate_gemma_kernel(
    const uint16_t *__restrict__ in, const uint16_t *__restrict__ sin, const uint16_t *__restrict__ cos, int rows, int q_hidden,
    int kv_hidden, int head_dim, int rope_dim, float eps, const uint16_t *__restrict__ qw, const uint16_t *__restrict__ kw,
    uint16_t *__restrict__ q, uint16_t *__restrict__ k, uint16_t *__restrict__ v, uint16_t *__restrict__ gate)
{
    const int lane = threadIdx.x & 63;
    const int q_heads = q_hidden / head_dim, kv_heads = kv_hidden / head_dim;
    const int items = q_heads + 2 * kv_heads;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long row = wid / items;
    const int it = (int)(wid % items);
    if (row >= rows) return;
    const long long total = 2ll * q_hidden + 2ll * kv_hidden;
    const uint16_t *rin = in + row * total;
    if (it >= q_heads + kv_heads) {                          // V: plain copy
        const int h = it - q_heads - kv_heads;
        const uint16_t *src = rin + 2ll * q_hidden + kv_hidden + (long long)h * head_dim;
        uint16_t *dst = v + row * (long long)kv_hidden + (long long)h * head_dim;
        for (int i = lane * 8; i < head_dim; i += 512) *(u32x4 *)(dst + i) = *(const u32x4 *)(src + i);
        return;
    }
    const bool is_q = it < q_heads;
    const uint16_t *src = is_q ? rin + (long long)it * 2 * head_dim : rin + 2ll * q_hidden + (long long)(it - q_heads) * head_dim;
    uint16_t *dst = is_q ? q + row * (long long)q_hidden + (long long)it * head_dim
                         : k + row * (long long)kv_hidden + (long long)(it - q_heads) * head_dim;
    const uint16_t *wt = is_q ? qw : kw;
    if (is_q) {                                              // the head's gate: copied
        const uint16_t *gs = src + head_dim;
        uint16_t *gd = gate + row * (long long)q_hidden + (long long)it * head_dim;
        for (int i = lane * 8; i < head_dim; i += 512) *(u32x4 *)(gd + i) = *(const u32x4 *)(gs + i);
    }
    // head_dim <= 2048: a lane holds up to 4 chunks of 8 elements
    constexpr int kMaxChunks = 4;
    float x[kMaxChunks][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < head_dim) {
            unpack8<BF16>(*(const u32x4 *)(src + i), x[c]);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += x[c][e] * x[c][e];
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)head_dim + eps);
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < head_dim) {
            float w[8];
            unpack8<BF16>(*(const u32x4 *)(wt + i), w);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[c][e] = (x[c][e] * rstd) * (w[e] + 1.0f);
        }
    }
    // RoPE: the partner of element p is p +- rope_dim / 2, i.e. a register of another lane: the normalised head is staged in a wave-private
    // LDS row (LDS operations of one wave complete in order: no workgroup barrier)
    __shared__ float stage[4][2048];
    float *sw = stage[threadIdx.x >> 6];
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < head_dim) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sw[i + e] = x[c][e];
        }
    }
    __builtin_amdgcn_wave_barrier();
    const int half = rope_dim >> 1;
    const uint16_t *sr = sin + row * (long long)rope_dim, *cr = cos + row * (long long)rope_dim;
#pragma unroll
    for (int c = 0; c < kMaxChunks; ++c) {
        const int i = (c * 64 + lane) * 8;
        if (i < head_dim) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int pidx = i + e;
                if (pidx < half) o[e] = (-sw[pidx + half]) * ld16<BF16>(sr[pidx]) + x[c][e] * ld16<BF16>(cr[pidx]);
                else if (pidx < rope_dim) o[e] = sw[pidx - half] * ld16<BF16>(sr[pidx]) + x[c][e] * ld16<BF16>(cr[pidx]);
                else o[e] = x[c][e];
            }
            *(u32x4 *)(dst + i) = pack8<BF16>(o);
        }
    }
}

// Vectorised variant: a head is spread over head_dim/8 lanes holding 8 consecutive elements each (16-B loads / stores),
// so a wave handles 64*8/head_dim heads.  The RoPE partner (p +- rope_dim/2) lives rope_dim/16 lanes away and is fetched
// with one shuffle per element; the RMS reduction is an xor-shuffle tree inside the head's lane group.
// Needs rope_dim % 16 == 0 (neox) or % 8 == 0 (interleaved); otherwise the scalar kernel above runs.
// Each wave handles kVecUnroll groups of 64*8/head_dim heads; ALL loads of all groups (x, norm weight / bias, sin, cos) are
// issued before any arithmetic, so a lane has up to 5 x kVecUnroll independent 16-byte loads in flight instead of a chain of
// three dependent round trips per kilobyte (the first version ran at 37 % of HBM).
// Round 2 (4096 x 8192, kernel time from rocprofv3): 38 -> 29 us = 4.6 TB/s: 32-bit index arithmetic instead of four 64-bit divisions
// per lane, v_cvt_pk_bf16_f32, DPP for the head reduction and the RoPE partner, two groups per wave (58 VGPRs, 8 waves per SIMD).
// Measured and dropped: a grid-stride loop over 2048 workgroups (31.8), NEOX / NORM as template parameters plus a copy path for
// all-V groups (31.5), sharing one sin / cos load between the groups of a wave (31.4; with the selection deferred to the use: 36).
// (round 4, same box: 1 item per wave 139 us at 16384 x 8192, 2 items 118, 4 items 118; non-temporal loads and stores 114)
constexpr int kVecUnroll = 2;

// MROPE (norm/split_qkv_rmsnorm_mrope.py:57-333; golden tests/python/sgl_kernel_npu/test_split_qkv_rmsnorm_mrope.py:7-110): `sin` is
// cos_sin [3, rows, rope_dim] -- per section (t, h, w) the first half of a row holds cos, the second sin --, `cos` unused; rotation
// offset o = p mod rope_dim / 2 takes its cos / sin from section h when (sections interleaved: o % 3 == 1 and o <= 3 sec1; contiguous:
// sec0 <= o < sec0 + sec1), from w when (o % 3 == 2 and o <= 3 sec2; contiguous: o >= sec0 + sec1), else from t.  Always rotate-half.
struct MropeSections {
    int sec0, sec1, sec2, interleaved;
    // mode 1 = position-indexed cache (norm/split_qkv_rmsnorm_rope_pos_cache_half_npu.py:25-230): `sin` is cos_sin_cache [max_seq, stride0]
    // (row = [cos half | sin half], element type cache_dtype: MI_DTYPE_BF16 / F16 / F32), the row of batch item b is pos[b] clamped to
    // [0, max_seq) (:76-79); cast_norm: the normalised value is rounded to the I/O dtype before the rotation (:118-121)
    int mode;
    const void *pos;
    int pos_is_i64, max_seq;
    long long stride0;
    int cache_dtype, cast_norm;
};
__device__ __forceinline__ int mrope_section_of(const MropeSections &m, int o)
{
    if (m.interleaved) {
        const int r = o % 3;
        if (r == 1 && o <= 3 * m.sec1) return 1;
        if (r == 2 && o <= 3 * m.sec2) return 2;
        return 0;
    }
    // offsets behind the three sections take cos = sin = 0 (the reference masks w to [t + h, t + h + w), :157-163)
    return o < m.sec0 ? 0 : (o < m.sec0 + m.sec1 ? 1 : (o < m.sec0 + m.sec1 + m.sec2 ? 2 : 3));
}
// FAST: the instance for the shape every Llama / Qwen-style caller has -- heads of 128, the whole head rotated, rotate-half, norm weights
// present, plain (not Gemma, not gated) -- with those facts as compile-time constants and NO per-lane branches: V heads run the same
// arithmetic as Q / K heads and keep their input words at the final select.  The general instance is VALU-bound (about 370 vector
// instructions per 16 B of every lane, a quarter of them moves and selects around the `normed` / `roped` branches: 4.5 TB/s).
template <bool BF16, bool MROPE, bool FAST = false>
__global__ __launch_bounds__(256) void split_qkv_rmsnorm_rope_vec_kernel(
    const uint16_t *__restrict__ qkv, const uint16_t *__restrict__ sin, const uint16_t *__restrict__ cos, int rows, int q_hidden,
    int kv_hidden, int head_dim_p, int rope_dim_p, int has_norm_p, float eps, const uint16_t *__restrict__ qw,
    const uint16_t *__restrict__ kw, const uint16_t *__restrict__ qb, const uint16_t *__restrict__ kb, int neox_p,
    uint16_t *__restrict__ q, uint16_t *__restrict__ k, uint16_t *__restrict__ v, uint16_t *__restrict__ gate, int gemma_p, MropeSections ms)
{
    // (FAST && MROPE: the same branch-free instance with the row's cos / sin pairs from the LDS table below -- the launcher guarantees the
    //  table form applies: at least 8 head-sized items per row, so a workgroup's heads span at most kMropeRows rows)
    const int head_dim = FAST ? 128 : head_dim_p, rope_dim = FAST ? 128 : rope_dim_p;
    const int has_norm = FAST ? 1 : has_norm_p, neox = FAST ? 1 : neox_p, gemma = FAST ? 0 : gemma_p;
    // gate != nullptr: the gated Gemma form (split_qkv_rmsnorm_rope.py:441-745) -- the row is q_heads pairs [q head | gate head], then K,
    // then V, i.e. still one head-sized item every head_dim elements; odd items of the first 2 q_heads are gates (copied like V).
    // gemma: the norm weight is w + 1 and the scale rsqrt(mean + eps) (:468, :499-503)
    const bool gated = FAST ? false : gate != nullptr;
    const int lane = threadIdx.x & 63;
    const int gl = head_dim >> 3;                      // lanes per head (8 .. 32), a power of two
    const int gl_shift = 31 - __builtin_clz(gl);
    const int heads_per_wave = 64 >> gl_shift;
    const int q_heads = q_hidden / head_dim, kv_heads = kv_hidden / head_dim;
    const int q_items = gated ? 2 * q_heads : q_heads;      // head-sized items in front of K
    const int heads_total = q_items + 2 * kv_heads;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int j = lane & (gl - 1);                     // chunk of 8 elements inside the head
    const long long total_hidden = (long long)q_items * head_dim + 2ll * kv_hidden;
    const int half = rope_dim >> 1;
    const bool roped = j * 8 < rope_dim;
    const u32x4 zero4 = u32x4{0, 0, 0, 0};

    long long row[kVecUnroll];
    int h[kVecUnroll];
    bool active[kVecUnroll];
    u32x4 xr[kVecUnroll], wr[kVecUnroll], br[kVecUnroll], sr[kVecUnroll], cr[kVecUnroll];
    float cvf[(MROPE && !FAST) ? kVecUnroll : 1][8], svf[(MROPE && !FAST) ? kVecUnroll : 1][8];      // MROPE (general instance): cos / sin of this lane's eight elements
    // MROPE with sections (mode 0): every head of a row rotates with the SAME selected cos / sin, and a workgroup's 4 x kVecUnroll x
    // heads_per_wave heads span one or two rows (up to kMropeRows) -- so the workgroup selects each row's rope_dim / 2 pairs ONCE into LDS
    // (one thread per (row, offset): two 2-byte loads from the offset's section) and every lane reads its eight pairs from there.  Read per
    // lane, the six 16-byte section vectors were 96 bytes of L2 traffic and ~100 selects for every 16 bytes of the row: 2.3 TB/s at 4096 x 8192.
    constexpr int kMropeRows = 8, kMropeHalf = 128;
    __shared__ float mrope_tab[(MROPE || FAST) ? kMropeRows * 2 * kMropeHalf : 1];
    bool mrope_lds = false;
    uint32_t mrope_row0 = 0;
    // FAST: the norm weights and biases of a 128-wide head are 16 chunks of 16 bytes each, the same for every head of the launch: staged once
    // per workgroup as well (threads 0..63: q weight | k weight | q bias | k bias), read from LDS per head instead of from global memory
    __shared__ u32x4 wtab[FAST ? 64 : 1];
    if (FAST) {
        if (threadIdx.x < 64) {
            const int which = threadIdx.x >> 4, c = threadIdx.x & 15;
            const uint16_t *srcp = which == 0 ? qw : (which == 1 ? kw : (which == 2 ? qb : kb));
            wtab[threadIdx.x] = srcp ? *(const u32x4 *)(srcp + c * 8) : zero4;
        }
    }
    if (FAST && !MROPE) {
        // the plain form's FAST instance takes its cos / sin from the same kind of table: entry o < rope_dim (= 128) of row r = cos / sin[row][o],
        // one thread per (row, o) -- instead of two 16-byte vectors per lane and head (the table forms of mrope / position cache ran FASTER than
        // the plain form until it got this: 24.7-25.4 against 26.4 us at 4096 x 8192)
        const uint32_t hg0 = (uint32_t)blockIdx.x * 4u * kVecUnroll * heads_per_wave, hcount = 4u * kVecUnroll * heads_per_wave;
        mrope_row0 = hg0 / (uint32_t)heads_total;
        const uint32_t row_last = min((hg0 + hcount - 1u) / (uint32_t)heads_total, (uint32_t)rows - 1u);
        const int nrows = min((int)row_last - (int)mrope_row0 + 1, kMropeRows);
        for (int idx = threadIdx.x; idx < nrows * rope_dim; idx += 256) {
            const int r = idx / rope_dim, o = idx - r * rope_dim;
            const long long src = (long long)(mrope_row0 + r) * rope_dim + o;
            mrope_tab[(r * 2 + 0) * kMropeHalf + o] = ld16<BF16>(cos[src]);
            mrope_tab[(r * 2 + 1) * kMropeHalf + o] = ld16<BF16>(sin[src]);
        }
        __syncthreads();
    }
    if (MROPE) {
        const uint32_t hg0 = (uint32_t)blockIdx.x * 4u * kVecUnroll * heads_per_wave, hcount = 4u * kVecUnroll * heads_per_wave;
        mrope_row0 = hg0 / (uint32_t)heads_total;
        const uint32_t row_last = min((hg0 + hcount - 1u) / (uint32_t)heads_total, (uint32_t)rows - 1u);
        const int nrows = (int)row_last - (int)mrope_row0 + 1;
        mrope_lds = half <= kMropeHalf && nrows >= 1 && nrows <= kMropeRows;      // workgroup-uniform
        if (mrope_lds) {
            const long long sec_stride = (long long)rows * rope_dim;
            for (int idx = threadIdx.x; idx < nrows * half; idx += 256) {
                const int r = idx / half, o = idx - r * half;
                float cvv = 0.f, svv = 0.f;
                if (ms.mode == 1) {                          // position-indexed cache: the row of this batch item's position
                    const long long rr = (long long)mrope_row0 + r;
                    long long pidx = ms.pos_is_i64 ? ((const long long *)ms.pos)[rr] : (long long)((const int *)ms.pos)[rr];
                    pidx = pidx < 0 ? 0 : (pidx > ms.max_seq - 1 ? ms.max_seq - 1 : pidx);
                    if (ms.cache_dtype == MI_DTYPE_F32) {
                        const float *base = (const float *)sin + pidx * ms.stride0 + o;
                        cvv = base[0], svv = base[half];
                    } else {
                        const uint16_t *base = sin + pidx * ms.stride0 + o;
                        cvv = ms.cache_dtype == MI_DTYPE_BF16 ? ld16<true>(base[0]) : ld16<false>(base[0]);
                        svv = ms.cache_dtype == MI_DTYPE_BF16 ? ld16<true>(base[half]) : ld16<false>(base[half]);
                    }
                } else if (const int sec = mrope_section_of(ms, o); sec < 3) {
                    const uint16_t *base = sin + sec * sec_stride + (long long)(mrope_row0 + r) * rope_dim + o;
                    cvv = ld16<BF16>(base[0]), svv = ld16<BF16>(base[half]);
                }
                mrope_tab[(r * 2 + 0) * kMropeHalf + o] = cvv;
                mrope_tab[(r * 2 + 1) * kMropeHalf + o] = svv;
            }
            __syncthreads();
        } else if (FAST) {
            __syncthreads();                                 // (wtab; the launcher only picks FAST where the table form applies)
        }
    }
#pragma unroll
    for (int u = 0; u < kVecUnroll; ++u) {
        // 32-bit index arithmetic (the launcher checks rows x heads < 2^31): four 64-bit divisions per lane were a large part of
        // this kernel's instruction stream
        const uint32_t hglobal = ((uint32_t)wid * kVecUnroll + u) * heads_per_wave + (lane >> gl_shift);
        const uint32_t r32 = hglobal / (uint32_t)heads_total;
        row[u] = r32;
        h[u] = (int)(hglobal - r32 * (uint32_t)heads_total);
        active[u] = row[u] < rows;
        const bool is_v = h[u] >= q_items + kv_heads || (gated && h[u] < q_items && (h[u] & 1)), is_q = h[u] < q_items;
        const bool normed = FAST ? active[u] : (active[u] && !is_v);      // FAST: V heads load (and compute) like the others
        xr[u] = active[u] ? *(const u32x4 *)(qkv + row[u] * total_hidden + (long long)h[u] * head_dim + j * 8) : zero4;
        if (FAST) {
            wr[u] = br[u] = zero4;          // (read from wtab where they are used)
        } else {
            wr[u] = (has_norm && normed) ? *(const u32x4 *)((is_q ? qw : kw) + j * 8) : zero4;
            br[u] = (has_norm && normed && qb) ? *(const u32x4 *)((is_q ? qb : kb) + j * 8) : zero4;
        }
        if (FAST) {
            sr[u] = zero4, cr[u] = zero4;       // (the cos / sin pairs are read from the LDS table where they are used: 32 registers less)
        } else if (MROPE) {
            sr[u] = zero4, cr[u] = zero4;
#pragma unroll
            for (int e = 0; e < 8; ++e) cvf[u][e] = 0.f, svf[u][e] = 0.f;
            if (normed && roped && ms.mode == 1 && !mrope_lds) {
                const int o0 = (j * 8) % half;
                long long pidx = ms.pos_is_i64 ? ((const long long *)ms.pos)[row[u]] : (long long)((const int *)ms.pos)[row[u]];
                pidx = pidx < 0 ? 0 : (pidx > ms.max_seq - 1 ? ms.max_seq - 1 : pidx);
                if (ms.cache_dtype == MI_DTYPE_F32) {
                    const float *base = (const float *)sin + pidx * ms.stride0 + o0;
#pragma unroll
                    for (int e = 0; e < 8; e += 4) {
                        const f32x4_t c4 = *(const f32x4_t *)(base + e), s4 = *(const f32x4_t *)(base + half + e);
#pragma unroll
                        for (int i = 0; i < 4; ++i) cvf[u][e + i] = c4[i], svf[u][e + i] = s4[i];
                    }
                } else {
                    const uint16_t *base = sin + pidx * ms.stride0 + o0;
                    const u32x4 c4 = *(const u32x4 *)base, s4 = *(const u32x4 *)(base + half);
                    if (ms.cache_dtype == MI_DTYPE_BF16) unpack8<true>(c4, cvf[u]), unpack8<true>(s4, svf[u]);
                    else unpack8<false>(c4, cvf[u]), unpack8<false>(s4, svf[u]);
                }
            } else if (normed && roped && mrope_lds) {
                const int o0 = (j * 8) % half;
                const float *tc = mrope_tab + ((int)(row[u] - mrope_row0) * 2) * kMropeHalf + o0;
#pragma unroll
                for (int e = 0; e < 8; e += 4) {
                    const f32x4_t c4 = *(const f32x4_t *)(tc + e), s4 = *(const f32x4_t *)(tc + kMropeHalf + e);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cvf[u][e + i] = c4[i], svf[u][e + i] = s4[i];
                }
            } else if (normed && roped) {
                const int o0 = (j * 8) % half;            // this lane's eight rotation offsets: the chunk does not straddle rope_dim / 2
                const long long sec_stride = (long long)rows * rope_dim;
                const uint16_t *base = sin + row[u] * (long long)rope_dim + o0;
                float c3[3][8], s3[3][8];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    unpack8<BF16>(*(const u32x4 *)(base + t * sec_stride), c3[t]);
                    unpack8<BF16>(*(const u32x4 *)(base + t * sec_stride + half), s3[t]);
                }
                const int r0 = o0 % 3;                   // one division per lane: (o0 + e) % 3 follows from it
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = o0 + e;
                    int sec;
                    if (ms.interleaved) {
                        const int r = (r0 + e) % 3;       // r0 + e < 10: folds to compares
                        sec = (r == 1 && o <= 3 * ms.sec1) ? 1 : ((r == 2 && o <= 3 * ms.sec2) ? 2 : 0);
                    } else {
                        sec = o < ms.sec0 ? 0 : (o < ms.sec0 + ms.sec1 ? 1 : (o < ms.sec0 + ms.sec1 + ms.sec2 ? 2 : 3));   // 3: behind the sections, cos = sin = 0
                    }
                    cvf[u][e] = sec == 0 ? c3[0][e] : (sec == 1 ? c3[1][e] : (sec == 2 ? c3[2][e] : 0.0f));
                    svf[u][e] = sec == 0 ? s3[0][e] : (sec == 1 ? s3[1][e] : (sec == 2 ? s3[2][e] : 0.0f));
                }
            }
        } else if (neox) {
            sr[u] = (normed && roped) ? *(const u32x4 *)(sin + row[u] * (long long)rope_dim + j * 8) : zero4;
            cr[u] = (normed && roped) ? *(const u32x4 *)(cos + row[u] * (long long)rope_dim + j * 8) : zero4;
        } else {
            // pairs (2i, 2i+1) use sin/cos[i]; this lane's 4 pairs are i = 4j .. 4j+3
            const u32x2 s2 = (normed && roped) ? *(const u32x2 *)(sin + row[u] * (long long)rope_dim + j * 4) : u32x2{0, 0};
            const u32x2 c2 = (normed && roped) ? *(const u32x2 *)(cos + row[u] * (long long)rope_dim + j * 4) : u32x2{0, 0};
            sr[u] = u32x4{s2[0], s2[1], 0, 0};
            cr[u] = u32x4{c2[0], c2[1], 0, 0};
        }
    }
#pragma unroll
    for (int u = 0; u < kVecUnroll; ++u) {
        const bool is_v = h[u] >= q_items + kv_heads || (gated && h[u] < q_items && (h[u] & 1)), is_q = h[u] < q_items;      // is_v: copied
        float x[8];
        unpack8<BF16>(xr[u], x);
        if (has_norm) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
            // sum over the head's gl lanes with DPP moves (one VALU instruction per step instead of a ds_bpermute round trip).  The
            // operand pairs are those of the xor tree (each step adds the two partial sums of sibling lane groups), so the result
            // is bit-identical to it.
            ss += dpp_f32<0xB1>(ss);                        // quad_perm [1,0,3,2]: lane ^ 1
            ss += dpp_f32<0x4E>(ss);                        // quad_perm [2,3,0,1]: lane ^ 2
            ss += dpp_f32<0x141>(ss);                       // row_half_mirror: the other quad of the 8-lane group
            if (gl >= 16) ss += dpp_f32<0x140>(ss);         // row_mirror: the other half of the 16-lane row
            if (gl == 32) ss += __shfl_xor(ss, 16, 64);
            if (FAST || (active[u] && !is_v)) {
                // rsqrt where the reference kernel says tl.rsqrt (gemma :500, position cache :101), 1 / sqrt where it says so (mrope :202)
                const float rstd = (gemma || (MROPE && ms.mode == 1)) ? rsqrtf(ss / (float)head_dim + eps) : 1.0f / sqrtf(ss / (float)head_dim + eps);
                float wv[8];
                if (FAST) unpack8<BF16>(wtab[(is_q ? 0 : 16) + j], wv);
                else unpack8<BF16>(wr[u], wv);
                if (gemma) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) wv[e] = wv[e] + 1.0f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (x[e] * rstd) * wv[e];
                if (qb) {
                    float bv[8];
                    if (FAST) unpack8<BF16>(wtab[(is_q ? 32 : 48) + j], bv);
                    else unpack8<BF16>(br[u], bv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = x[e] + bv[e];
                }
                if (MROPE && ms.cast_norm) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = ld16<BF16>(st16<BF16>(x[e]));
                }
            }
        }
        float o[8];
        if (neox) {
            const int dl = half >> 3;                      // partner distance in lanes
            const bool lower = (j * 8) < half;
            const int partner = lower ? lane + dl : lane - dl;
            float px[8];
            if (dl == 8) {                                 // rope_dim 128: the partner is lane ^ 8 = a rotation by 8 inside the 16-lane row
#pragma unroll
                for (int e = 0; e < 8; ++e) px[e] = dpp_f32<0x128>(x[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) px[e] = __shfl(x[e], partner & 63, 64);
            }
            if (FAST || (active[u] && !is_v && roped)) {
                float sv[8], cv[8];
                if (FAST) {
                    // entry of this lane's eight elements: mrope / position cache tables hold rope_dim / 2 pairs (offset p mod rope_dim / 2), the plain
                    // form's rope_dim pairs; inactive lanes read a row of the table, unused
                    const int o0 = MROPE ? (j * 8) % half : j * 8;
                    const int rr = min(max((int)(row[u] - mrope_row0), 0), kMropeRows - 1);
                    const float *tc = mrope_tab + (rr * 2) * kMropeHalf + o0;
#pragma unroll
                    for (int e = 0; e < 8; e += 4) {
                        const f32x4_t c4 = *(const f32x4_t *)(tc + e), s4 = *(const f32x4_t *)(tc + kMropeHalf + e);
#pragma unroll
                        for (int i = 0; i < 4; ++i) cv[e + i] = c4[i], sv[e + i] = s4[i];
                    }
                } else if (MROPE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sv[e] = svf[u][e], cv[e] = cvf[u][e];
                } else {
                    unpack8<BF16>(sr[u], sv);
                    unpack8<BF16>(cr[u], cv);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (lower ? -px[e] : px[e]) * sv[e] + x[e] * cv[e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = x[e];
            }
        } else {
            if (active[u] && !is_v && roped) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sv = ld16<BF16>((sr[u][i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
                    const float cv = ld16<BF16>((cr[u][i >> 1] >> (16 * (i & 1))) & 0xFFFFu);
                    const float x1 = x[2 * i], x2 = x[2 * i + 1];
                    o[2 * i] = (-x2) * sv + x1 * cv;
                    o[2 * i + 1] = x1 * sv + x2 * cv;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = x[e];
            }
        }
        if (!active[u]) continue;
        uint16_t *dst;
        if (is_q) dst = ((gated && (h[u] & 1)) ? gate : q) + row[u] * (long long)q_hidden + (long long)(gated ? h[u] >> 1 : h[u]) * head_dim;
        else if (h[u] < q_items + kv_heads) dst = k + row[u] * (long long)kv_hidden + (long long)(h[u] - q_items) * head_dim;
        else dst = v + row[u] * (long long)kv_hidden + (long long)(h[u] - q_items - kv_heads) * head_dim;
        *(u32x4 *)(dst + j * 8) = is_v ? xr[u] : pack8<BF16>(o);
    }
}

}  // namespace mi_sgl

using namespace mi_sgl;

static int launch_ok() { return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH; }

extern "C" int mi_swiglu_quant(const void *x, const void *group_list, int group_list_is_i64, int num_groups, int group_list_type,
                               int rows, int cols, int need_quant, int do_limit, float limit, int dtype, void *out, float *scale,
                               void *stream)
{
    if (rows < 0 || cols <= 0 || cols % 16 || cols / 2 > kSwigluItems * 512 || num_groups <= 0 ||
        (group_list_type != 0 && group_list_type != 1) || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!x || !group_list || !out || (need_quant && !scale)) return MI_SGL_EINVAL;
    const int blocks = (rows + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
    const int per_lane = (cols / 2 / 8 + 63) / 64;       // 16-byte items per lane
#define MI_LAUNCH_N(B, L, N)                                                                                                     \
    swiglu_quant_kernel<B, L, N><<<blocks, 256, 0, st>>>((const uint16_t *)x, group_list, num_groups, group_list_type, rows,       \
                                                         cols / 2, need_quant, do_limit, limit, out, scale)
#define MI_LAUNCH(B, L)                                                                                                          \
    do {                                                                                                                         \
        if (per_lane <= 1) MI_LAUNCH_N(B, L, 1);                                                                                 \
        else if (per_lane <= 2) MI_LAUNCH_N(B, L, 2);                                                                            \
        else if (per_lane <= 4) MI_LAUNCH_N(B, L, 4);                                                                            \
        else MI_LAUNCH_N(B, L, 8);                                                                                               \
    } while (0)
    if (dtype == MI_DTYPE_BF16) { if (group_list_is_i64) MI_LAUNCH(true, true); else MI_LAUNCH(true, false); }
    else { if (group_list_is_i64) MI_LAUNCH(false, true); else MI_LAUNCH(false, false); }
#undef MI_LAUNCH
#undef MI_LAUNCH_N
    return launch_ok();
}

extern "C" int mi_add_rmsnorm_bias(const void *input, const void *residual, const void *weight, const void *bias, float eps,
                                   const void *quant_scale, const void *quant_offset, int gemma, int rows, int hidden,
                                   int64_t input_row_stride, int dtype, void *out, void *out2, void *stream)
{
    if (rows < 0 || hidden <= 0 || hidden % 8 || hidden > kNormItems * 2048 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) ||
        ((quant_scale == nullptr) != (quant_offset == nullptr)) || input_row_stride % 8)
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!input || !weight || !out) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI_DTYPE_BF16)
        add_rmsnorm_kernel<true><<<rows, 256, 0, st>>>((const uint16_t *)input, (const uint16_t *)residual, (const uint16_t *)weight,
                                                      (const uint16_t *)bias, (const uint16_t *)quant_scale,
                                                      (const uint16_t *)quant_offset, eps, gemma, hidden, input_row_stride, out,
                                                      (uint16_t *)out2);
    else
        add_rmsnorm_kernel<false><<<rows, 256, 0, st>>>((const uint16_t *)input, (const uint16_t *)residual, (const uint16_t *)weight,
                                                       (const uint16_t *)bias, (const uint16_t *)quant_scale,
                                                       (const uint16_t *)quant_offset, eps, gemma, hidden, input_row_stride, out,
                                                       (uint16_t *)out2);
    return launch_ok();
}

// ------------------------------------------------------------------------------------------------
// RoPE on q [T, Hq, D] and the (few) key heads k [T, Hk, D] with one cos|sin row per token
// (reference norm/fused_rope_qk_mqa.py:6-160: cos = cos_sin[t, :R/2], sin = cos_sin[t, R/2:R]).  One wave per (token, head).
// Arithmetic follows the reference kernel / its test golden op by op in the I/O dtype: o1 = r(r(x1*c) - r(x2*s)),
// o2 = r(r(x1*s) + r(x2*c)) with r = round to the I/O dtype (products of two 16-bit floats are exact in fp32).
// ------------------------------------------------------------------------------------------------
namespace {
template <bool BF16>
__global__ __launch_bounds__(256) void rope_qk_mqa_kernel(const uint16_t *__restrict__ q, const uint16_t *__restrict__ k,
                                                         const uint16_t *__restrict__ cos_sin, int T, int Hq, int Hk, int D, int R,
                                                         int neox, long long q_st, long long q_sh, long long k_st, long long k_sh,
                                                         long long cs_st, uint16_t *__restrict__ oq, uint16_t *__restrict__ ok)
{
    const int lane = threadIdx.x & 63;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int H = Hq + Hk;
    const long long t = wid / H;
    const int h = (int)(wid % H);
    if (t >= T) return;
    const bool is_q = h < Hq;
    const uint16_t *src = is_q ? q + t * q_st + (long long)h * q_sh : k + t * k_st + (long long)(h - Hq) * k_sh;
    uint16_t *dst = is_q ? oq + (t * Hq + h) * (long long)D : ok + (t * Hk + (h - Hq)) * (long long)D;
    const uint16_t *cs = cos_sin + t * cs_st;
    const int half = R >> 1;
    auto r = [](float f) -> float { return ld16<BF16>(st16<BF16>(f)); };
    for (int i = lane; i < half; i += 64) {
        const int ie = neox ? i : 2 * i, io = neox ? i + half : 2 * i + 1;
        const float x1 = ld16<BF16>(src[ie]), x2 = ld16<BF16>(src[io]);
        const float c = ld16<BF16>(cs[i]), sn = ld16<BF16>(cs[half + i]);
        dst[ie] = (uint16_t)st16<BF16>(r(x1 * c) - r(x2 * sn));
        dst[io] = (uint16_t)st16<BF16>(r(x1 * sn) + r(x2 * c));
    }
    for (int i = R + lane; i < D; i += 64) dst[i] = src[i];
}
// Vectorised form: a lane owns 8 consecutive elements (16 bytes) of a head; a head is D / 8 lanes, heads and tokens are laid end to end over
// the grid.  Rotate-half: the partner chunk (R / 2 elements away) is loaded by the lane itself -- an L1 / L2 hit of the line its neighbour
// lane reads -- so no cross-lane traffic and any D; interleaved pairs live inside the lane's own chunk.  cos / sin: 16 (8) bytes each per
// lane from the token's row.  Same op-by-op arithmetic as the scalar kernel above (which keeps serving D % 8 != 0, R % 16 != 0 or unaligned
// strides): bit-identical outputs.  The scalar kernel moved 2 bytes per lane and instruction: 1.6-2.3 TB/s at 4096 tokens x 128 heads.
template <bool BF16>
__global__ __launch_bounds__(256) void rope_qk_mqa_vec_kernel(const uint16_t *__restrict__ q, const uint16_t *__restrict__ k,
                                                             const uint16_t *__restrict__ cos_sin, long long lanes_total, int Hq, int Hk, int D,
                                                             int R, int neox, long long q_st, long long q_sh, long long k_st, long long k_sh,
                                                             long long cs_st, uint16_t *__restrict__ oq, uint16_t *__restrict__ ok)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= lanes_total) return;
    // lanes of a head: rotate-half -- one per PAIR of chunks (elements e0 .. e0 + 7 and their partners R / 2 further: both halves of the
    // rotation from two loads, nothing read twice), then one per copied chunk; interleaved -- one per chunk
    const int half = R >> 1;
    const uint32_t rot_lanes = (uint32_t)(neox ? half : R) >> 3, lph = rot_lanes + ((uint32_t)(D - R) >> 3), H = (uint32_t)(Hq + Hk);
    const uint32_t hg = (uint32_t)(gid / lph), j = (uint32_t)(gid - (long long)hg * lph);      // head (token-major), lane inside the head
    const uint32_t t = hg / H, h = hg - t * H;
    const bool is_q = h < (uint32_t)Hq;
    const uint16_t *src = is_q ? q + (long long)t * q_st + (long long)h * q_sh : k + (long long)t * k_st + (long long)(h - Hq) * k_sh;
    uint16_t *dst = is_q ? oq + ((long long)t * Hq + h) * (long long)D : ok + ((long long)t * Hk + (h - Hq)) * (long long)D;
    if (j >= rot_lanes) {                                 // behind the rotated part: copied
        const int e0 = R + (int)(j - rot_lanes) * 8;
        *(u32x4 *)(dst + e0) = *(const u32x4 *)(src + e0);
        return;
    }
    const int e0 = (int)j * 8;
    const uint16_t *cs = cos_sin + (long long)t * cs_st;
    auto r = [](float f) -> float { return ld16<BF16>(st16<BF16>(f)); };
    if (neox) {
        const u32x4 x1v = *(const u32x4 *)(src + e0), x2v = *(const u32x4 *)(src + e0 + half);
        const u32x4 cv = *(const u32x4 *)(cs + e0), sv = *(const u32x4 *)(cs + half + e0);
        float x1[8], x2[8], c[8], sn[8], o1[8], o2[8];
        unpack8<BF16>(x1v, x1), unpack8<BF16>(x2v, x2), unpack8<BF16>(cv, c), unpack8<BF16>(sv, sn);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o1[e] = r(x1[e] * c[e]) - r(x2[e] * sn[e]);
            o2[e] = r(x1[e] * sn[e]) + r(x2[e] * c[e]);
        }
        *(u32x4 *)(dst + e0) = pack8<BF16>(o1);
        *(u32x4 *)(dst + e0 + half) = pack8<BF16>(o2);
    } else {
        float x[8], o[8];
        unpack8<BF16>(*(const u32x4 *)(src + e0), x);
        const u32x2 cw = *(const u32x2 *)(cs + (e0 >> 1)), sw = *(const u32x2 *)(cs + half + (e0 >> 1));
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float c = ld16<BF16>((cw[p >> 1] >> (16 * (p & 1))) & 0xFFFFu), sn = ld16<BF16>((sw[p >> 1] >> (16 * (p & 1))) & 0xFFFFu);
            const float x1 = x[2 * p], x2 = x[2 * p + 1];
            o[2 * p] = r(x1 * c) - r(x2 * sn);
            o[2 * p + 1] = r(x1 * sn) + r(x2 * c);
        }
        *(u32x4 *)(dst + e0) = pack8<BF16>(o);
    }
}
}  // namespace

extern "C" int mi_rope_qk_mqa(const void *q, const void *k, const void *cos_sin, int tokens, int q_heads, int k_heads, int head_dim,
                              int rope_dim, int neox, int64_t q_stride_t, int64_t q_stride_h, int64_t k_stride_t, int64_t k_stride_h,
                              int64_t cs_stride_t, int dtype, void *out_q, void *out_k, void *stream)
{
    if (tokens < 0 || q_heads <= 0 || k_heads <= 0 || head_dim <= 0 || rope_dim <= 0 || rope_dim > head_dim || rope_dim % 2 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16))
        return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!q || !k || !cos_sin || !out_q || !out_k) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    {
        // 16-byte lanes when the shapes allow it: whole chunks (D % 8), the rotated part ending on a chunk and -- rotate-half -- its halves too,
        // every row 16-byte aligned
        const bool aligned = !(((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos_sin | (uintptr_t)out_q | (uintptr_t)out_k) & 15) &&
                             !((q_stride_t | q_stride_h | k_stride_t | k_stride_h | cs_stride_t) & 7);
        const long long chunks = (long long)tokens * (q_heads + k_heads) * (((neox ? rope_dim / 2 : rope_dim) + head_dim - rope_dim) / 8);      // lanes
        static const bool allow_vec = !(getenv("MI_ROPE_QK_VEC") && atoi(getenv("MI_ROPE_QK_VEC")) == 0);
        if (allow_vec && aligned && head_dim % 8 == 0 && rope_dim % (neox ? 16 : 8) == 0 && (long long)tokens * (q_heads + k_heads) < (1ll << 31) &&
            chunks / 256 < (1ll << 31) - 1) {
            const int vblocks = (int)((chunks + 255) / 256);
            if (dtype == MI_DTYPE_BF16)
                rope_qk_mqa_vec_kernel<true><<<vblocks, 256, 0, st>>>((const uint16_t *)q, (const uint16_t *)k, (const uint16_t *)cos_sin, chunks, q_heads,
                                                                    k_heads, head_dim, rope_dim, neox, q_stride_t, q_stride_h, k_stride_t, k_stride_h,
                                                                    cs_stride_t, (uint16_t *)out_q, (uint16_t *)out_k);
            else
                rope_qk_mqa_vec_kernel<false><<<vblocks, 256, 0, st>>>((const uint16_t *)q, (const uint16_t *)k, (const uint16_t *)cos_sin, chunks, q_heads,
                                                                     k_heads, head_dim, rope_dim, neox, q_stride_t, q_stride_h, k_stride_t, k_stride_h,
                                                                     cs_stride_t, (uint16_t *)out_q, (uint16_t *)out_k);
            return launch_ok();
        }
    }
    const long long waves = (long long)tokens * (q_heads + k_heads);
    const int blocks = (int)((waves + 3) / 4);
    if (dtype == MI_DTYPE_BF16)
        rope_qk_mqa_kernel<true><<<blocks, 256, 0, st>>>((const uint16_t *)q, (const uint16_t *)k, (const uint16_t *)cos_sin, tokens,
                                                        q_heads, k_heads, head_dim, rope_dim, neox, q_stride_t, q_stride_h,
                                                        k_stride_t, k_stride_h, cs_stride_t, (uint16_t *)out_q, (uint16_t *)out_k);
    else
        rope_qk_mqa_kernel<false><<<blocks, 256, 0, st>>>((const uint16_t *)q, (const uint16_t *)k, (const uint16_t *)cos_sin, tokens,
                                                         q_heads, k_heads, head_dim, rope_dim, neox, q_stride_t, q_stride_h,
                                                         k_stride_t, k_stride_h, cs_stride_t, (uint16_t *)out_q, (uint16_t *)out_k);
    return launch_ok();
}

extern "C" int mi_split_qkv_rmsnorm_rope(const void *qkv, const void *sin, const void *cos, int rows, int q_hidden, int kv_hidden,
                                         int head_dim, int rope_dim, int has_norm, float eps, const void *q_weight,
                                         const void *k_weight, const void *q_bias, const void *k_bias, int neox, int dtype,
                                         void *q, void *k, void *v, void *stream)
{
    if (rows < 0 || head_dim <= 0 || (head_dim & (head_dim - 1)) || q_hidden % head_dim || kv_hidden % head_dim ||
        kv_hidden <= 0 || q_hidden % kv_hidden || rope_dim <= 0 || rope_dim > head_dim || rope_dim % 2 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!qkv || !sin || !cos || !q || !k || !v || (has_norm && (!q_weight || !k_weight)) || ((q_bias == nullptr) != (k_bias == nullptr)))
        return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int heads_total = (q_hidden + 2 * kv_hidden) / head_dim;
    const bool vec_ok = head_dim >= 64 && head_dim <= 256 && (neox ? rope_dim % 16 == 0 : rope_dim % 8 == 0) &&
                        (long long)rows * heads_total < (1ll << 31) - 4096;
    if (vec_ok) {
        const long long heads = (long long)rows * heads_total;
        const int hpw = 64 / (head_dim / 8);
        const long long waves = ((heads + hpw - 1) / hpw + kVecUnroll - 1) / kVecUnroll;
        const int blocks = (int)((waves + 3) / 4);
#define MI_VEC(B)                                                                                                                   \
    split_qkv_rmsnorm_rope_vec_kernel<B, false><<<blocks, 256, 0, st>>>(                                                            \
        (const uint16_t *)qkv, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden, kv_hidden, head_dim, rope_dim, has_norm, eps, \
        (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)q_bias, (const uint16_t *)k_bias, neox, (uint16_t *)q,   \
        (uint16_t *)k, (uint16_t *)v, (uint16_t *)nullptr, 0, MropeSections{0, 0, 0, 0, 0, nullptr, 0, 0, 0, 0, 0})
        // the common shape -- heads of 128 rotated whole, rotate-half, norm weights -- has its own branch-free instance
        const bool fast = head_dim == 128 && rope_dim == 128 && neox && has_norm && heads_total >= 8;      // (>= 8 items per row: the cos / sin table form)
#define MI_VEC_FAST(B)                                                                                                              \
    split_qkv_rmsnorm_rope_vec_kernel<B, false, true><<<blocks, 256, 0, st>>>(                                                      \
        (const uint16_t *)qkv, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden, kv_hidden, head_dim, rope_dim, has_norm, eps, \
        (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)q_bias, (const uint16_t *)k_bias, neox, (uint16_t *)q,   \
        (uint16_t *)k, (uint16_t *)v, (uint16_t *)nullptr, 0, MropeSections{0, 0, 0, 0, 0, nullptr, 0, 0, 0, 0, 0})
        static const bool allow_fast = !(getenv("MI_SPLIT_QKV_FAST") && atoi(getenv("MI_SPLIT_QKV_FAST")) == 0);
        if (fast && allow_fast) { if (dtype == MI_DTYPE_BF16) MI_VEC_FAST(true); else MI_VEC_FAST(false); }
        else if (dtype == MI_DTYPE_BF16) MI_VEC(true); else MI_VEC(false);
#undef MI_VEC_FAST
#undef MI_VEC
        return launch_ok();
    }
    const long long waves = (long long)rows * heads_total;
    const int blocks = (int)((waves + 3) / 4);
    if (dtype == MI_DTYPE_BF16)
        split_qkv_rmsnorm_rope_kernel<true><<<blocks, 256, 0, st>>>(
            (const uint16_t *)qkv, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden, kv_hidden, head_dim, rope_dim, has_norm,
            eps, (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)q_bias, (const uint16_t *)k_bias, neox,
            (uint16_t *)q, (uint16_t *)k, (uint16_t *)v);
    else
        split_qkv_rmsnorm_rope_kernel<false><<<blocks, 256, 0, st>>>(
            (const uint16_t *)qkv, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden, kv_hidden, head_dim, rope_dim, has_norm,
            eps, (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)q_bias, (const uint16_t *)k_bias, neox,
            (uint16_t *)q, (uint16_t *)k, (uint16_t *)v);
    return launch_ok();
}

extern "C" int mi_split_qkvgate_gemma_rmsnorm_rope(const void *input, const void *sin, const void *cos, int rows, int q_hidden, int kv_hidden,
                                                   int head_dim, int rope_dim, float eps, const void *q_weight, const void *k_weight,
                                                   int dtype, void *q, void *k, void *v, void *gate, void *stream)
{
    if (rows < 0 || head_dim < 8 || head_dim > 2048 || (head_dim & (head_dim - 1)) || q_hidden <= 0 || q_hidden % head_dim ||
        kv_hidden <= 0 || kv_hidden % head_dim || q_hidden % kv_hidden || rope_dim <= 0 || rope_dim > head_dim || rope_dim % 2 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!input || !sin || !cos || !q_weight || !k_weight || !q || !k || !v || !gate) return MI_SGL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // the vectorised kernel of split_qkv_rmsnorm_rope serves the common shapes (a head spread over head_dim / 8 lanes); the wave-per-item
    // kernel above is the general form
    const int items_total = (2 * q_hidden + 2 * kv_hidden) / head_dim;
    if (head_dim >= 64 && head_dim <= 256 && rope_dim % 16 == 0 && (long long)rows * items_total < (1ll << 31) - 4096) {
        const long long heads = (long long)rows * items_total;
        const int hpw = 64 / (head_dim / 8);
        const long long vwaves = ((heads + hpw - 1) / hpw + kVecUnroll - 1) / kVecUnroll;
        const int vblocks = (int)((vwaves + 3) / 4);
#define MI_GVEC(B)                                                                                                                   \
    split_qkv_rmsnorm_rope_vec_kernel<B, false><<<vblocks, 256, 0, st>>>(                                                            \
        (const uint16_t *)input, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden, kv_hidden, head_dim, rope_dim, 1, eps,  \
        (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)nullptr, (const uint16_t *)nullptr, 1, (uint16_t *)q,  \
        (uint16_t *)k, (uint16_t *)v, (uint16_t *)gate, 1, MropeSections{0, 0, 0, 0, 0, nullptr, 0, 0, 0, 0, 0})
        if (dtype == MI_DTYPE_BF16) MI_GVEC(true); else MI_GVEC(false);
#undef MI_GVEC
        return launch_ok();
    }
    const long long waves = (long long)rows * ((q_hidden + 2 * kv_hidden) / head_dim);
    if (waves > (1ll << 31) - 8) return MI_SGL_EINVAL;
    const int blocks = (int)((waves + 3) / 4);
    if (dtype == MI_DTYPE_BF16)
        split_qkvgate_gemma_kernel<true><<<blocks, 256, 0, st>>>((const uint16_t *)input, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden,
                                                                 kv_hidden, head_dim, rope_dim, eps, (const uint16_t *)q_weight,
                                                                 (const uint16_t *)k_weight, (uint16_t *)q, (uint16_t *)k, (uint16_t *)v,
                                                                 (uint16_t *)gate);
    else
        split_qkvgate_gemma_kernel<false><<<blocks, 256, 0, st>>>((const uint16_t *)input, (const uint16_t *)sin, (const uint16_t *)cos, rows, q_hidden,
                                                                  kv_hidden, head_dim, rope_dim, eps, (const uint16_t *)q_weight,
                                                                  (const uint16_t *)k_weight, (uint16_t *)q, (uint16_t *)k, (uint16_t *)v,
                                                                  (uint16_t *)gate);
    return launch_ok();
}

extern "C" int mi_split_qkv_rmsnorm_mrope(const void *qkv, const void *cos_sin, int rows, int q_hidden, int kv_hidden, int head_dim, int rope_dim,
                                          float eps, const void *q_weight, const void *k_weight, const void *q_bias, const void *k_bias, int sec_t,
                                          int sec_h, int sec_w, int sections_interleaved, int dtype, void *q, void *k, void *v, void *gate,
                                          void *stream)
{
    if (rows < 0 || head_dim < 64 || head_dim > 256 || (head_dim & (head_dim - 1)) || q_hidden <= 0 || q_hidden % head_dim || kv_hidden <= 0 ||
        kv_hidden % head_dim || rope_dim <= 0 || rope_dim > head_dim || rope_dim % 16 || sec_t < 0 || sec_h < 0 || sec_w < 0 ||
        (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!qkv || !cos_sin || !q_weight || !k_weight || !q || !k || !v || ((q_bias == nullptr) != (k_bias == nullptr))) return MI_SGL_EINVAL;
    const int items_total = ((gate ? 2 : 1) * q_hidden + 2 * kv_hidden) / head_dim;
    if ((long long)rows * items_total >= (1ll << 31) - 4096) return MI_SGL_EINVAL;
    const long long heads = (long long)rows * items_total;
    const int hpw = 64 / (head_dim / 8);
    const long long vwaves = ((heads + hpw - 1) / hpw + kVecUnroll - 1) / kVecUnroll;
    const int vblocks = (int)((vwaves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    const MropeSections ms{sec_t, sec_h, sec_w, sections_interleaved ? 1 : 0, 0, nullptr, 0, 0, 0, 0, 0};
#define MI_MVEC(B, F)                                                                                                                  \
    split_qkv_rmsnorm_rope_vec_kernel<B, true, F><<<vblocks, 256, 0, st>>>(                                                            \
        (const uint16_t *)qkv, (const uint16_t *)cos_sin, (const uint16_t *)nullptr, rows, q_hidden, kv_hidden, head_dim, rope_dim, 1, eps, \
        (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)q_bias, (const uint16_t *)k_bias, 1, (uint16_t *)q,    \
        (uint16_t *)k, (uint16_t *)v, (uint16_t *)gate, 0, ms)
    // heads of 128 rotated whole, plain (not gated), enough items per row for the LDS table: the branch-free instance (MI_SPLIT_QKV_FAST=0: the general one)
    static const bool allow_fast = !(getenv("MI_SPLIT_QKV_FAST") && atoi(getenv("MI_SPLIT_QKV_FAST")) == 0);
    const bool fast = allow_fast && head_dim == 128 && rope_dim == 128 && !gate && items_total >= 8;
    if (fast) { if (dtype == MI_DTYPE_BF16) MI_MVEC(true, true); else MI_MVEC(false, true); }
    else if (dtype == MI_DTYPE_BF16) MI_MVEC(true, false); else MI_MVEC(false, false);
#undef MI_MVEC
    return launch_ok();
}

extern "C" int mi_split_qkv_rmsnorm_rope_pos_cache(const void *qkv, const void *positions, int pos_is_i64, const void *cos_sin_cache, int cache_dtype,
                                                   int max_seq, long long cache_stride0, int rows, int q_hidden, int kv_hidden, int head_dim, int rope_dim,
                                                   int has_norm, float eps, const void *q_weight, const void *k_weight, const void *q_bias,
                                                   const void *k_bias, int cast_norm, int dtype, void *q, void *k, void *v, void *stream)
{
    if (rows < 0 || head_dim < 64 || head_dim > 256 || (head_dim & (head_dim - 1)) || q_hidden <= 0 || q_hidden % head_dim || kv_hidden <= 0 ||
        kv_hidden % head_dim || q_hidden % kv_hidden || rope_dim <= 0 || rope_dim > head_dim || rope_dim % 16 || max_seq < 1 ||
        cache_stride0 < rope_dim || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) ||
        (cache_dtype != MI_DTYPE_BF16 && cache_dtype != MI_DTYPE_F16 && cache_dtype != MI_DTYPE_F32) ||
        (cache_stride0 % (cache_dtype == MI_DTYPE_F32 ? 4 : 8)))
        return MI_SGL_EINVAL;
    if (rows == 0) return MI_SGL_OK;
    if (!qkv || !positions || !cos_sin_cache || !q || !k || !v || (has_norm && (!q_weight || !k_weight)) || ((q_bias == nullptr) != (k_bias == nullptr)) ||
        (!has_norm && q_bias))
        return MI_SGL_EINVAL;
    const int items_total = (q_hidden + 2 * kv_hidden) / head_dim;
    if ((long long)rows * items_total >= (1ll << 31) - 4096) return MI_SGL_EINVAL;
    const long long heads = (long long)rows * items_total;
    const int hpw = 64 / (head_dim / 8);
    const long long vwaves = ((heads + hpw - 1) / hpw + kVecUnroll - 1) / kVecUnroll;
    const int vblocks = (int)((vwaves + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    const MropeSections ms{0, 0, 0, 0, 1, positions, pos_is_i64 ? 1 : 0, max_seq, cache_stride0, cache_dtype, cast_norm ? 1 : 0};
#define MI_PVEC(B, F)                                                                                                                  \
    split_qkv_rmsnorm_rope_vec_kernel<B, true, F><<<vblocks, 256, 0, st>>>(                                                            \
        (const uint16_t *)qkv, (const uint16_t *)cos_sin_cache, (const uint16_t *)nullptr, rows, q_hidden, kv_hidden, head_dim, rope_dim, \
        has_norm, eps, (const uint16_t *)q_weight, (const uint16_t *)k_weight, (const uint16_t *)q_bias, (const uint16_t *)k_bias, 1,   \
        (uint16_t *)q, (uint16_t *)k, (uint16_t *)v, (uint16_t *)nullptr, 0, ms)
    static const bool allow_fast = !(getenv("MI_SPLIT_QKV_FAST") && atoi(getenv("MI_SPLIT_QKV_FAST")) == 0);
    const bool fast = allow_fast && head_dim == 128 && rope_dim == 128 && has_norm && items_total >= 8;
    if (fast) { if (dtype == MI_DTYPE_BF16) MI_PVEC(true, true); else MI_PVEC(false, true); }
    else if (dtype == MI_DTYPE_BF16) MI_PVEC(true, false); else MI_PVEC(false, false);
#undef MI_PVEC
    return launch_ok();
}
