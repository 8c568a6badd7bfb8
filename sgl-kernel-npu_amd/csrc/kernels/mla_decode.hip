// Paged MLA decode attention for gfx950 (MI355X), bf16 / fp16.
// Replaces the reference Triton kernel _paged_mla_fwd_kernel / decode_mla
// (python/sgl_kernel_npu/sgl_kernel_npu/attention/decode_attention.py:5-230): same math (scores in fp32, online softmax,
// P rounded to the KV dtype, V aliases K_nope, out = acc / l), different schedule.
//
// MI355X design (BASELINE C4: B=128, 128 q-heads sharing one latent KV head, D = 512 + 64, seqlen 4096)
//  * arithmetic intensity ~242 FLOP/B sits under the HBM/MFMA ridge (~400): the KV stream must be read from HBM exactly
//    once, so ONE workgroup serves all (up to 128) heads of a sequence: 8 waves x 16 heads (the reference launches one
//    program per 16 heads and re-reads KV 8 times);
//  * KV tiles of 64 keys go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR
//    staging), double buffered (2 x 74.75 KB of the 160 KB LDS), ONE barrier per tile; any page_size works because every
//    key row is addressed through the block table individually;
//  * both GEMMs run transposed so nothing is shuffled between them: S^T[key, head] = K · Q^T (A = K rows from LDS via
//    ds_read_b128, B = Q^T fragments resident in 72 VGPRs) and O^T[d, head] += V^T · P^T (A = V^T through the LDS
//    transpose read ds_read_b64_tr_b16, B = P^T which IS the S^T accumulator layout, packed to bf16 in place);
//    v_mfma_f32_16x16x32_{bf16,f16}; softmax statistics are one value per lane (lane = head);
//  * nope rows are padded to 1040 B (conflict-free ds_read_b128 across the 16 keys of an MFMA operand); the 128-B rope
//    rows are XOR-swizzled by choosing which global chunk each LDS-DMA lane fetches;
//  * flash-decoding split over the KV range fills the 256 CUs when batch < 256 (C4: 2 splits -> 256 workgroups); a small
//    kernel merges the (m, l, O) partials.
// Algorithmic HBM bytes per launch: sum_b len_b * 576 * 2 (KV once) + B*Hq*(576+512)*2 (Q, out); FLOPs
// sum_b Hq * len_b * (576+512) * 2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mi_sgl_kernels.h"

namespace mi_sgl {

constexpr int kDN = 512, kDR = 64, kTile = 64;
constexpr int kNopeStride = kDN * 2 + 16;          // bytes per key row in LDS
constexpr int kRopeStride = kDR * 2;               // 128 B, swizzled
constexpr int kBufBytes = kTile * kNopeStride + kTile * kRopeStride;   // 74752
constexpr int kMaxWaves = 8;
constexpr int kHeadsPerBlock = kMaxWaves * 16;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct MlaParams {
    const uint16_t *q, *k_nope, *k_rope;
    uint16_t *out;
    const int32_t *seq_lens, *block_table;
    float *ws_o;      // [B][Hq][S][512] fp32 partial (unnormalised) outputs
    float *ws_ml;     // [B][Hq][S][2]   running max, running sum
    int batch, q_heads, kv_heads, group, page_size, bt_stride, num_splits;
    int64_t q_sb, q_sh, kn_sblk, kn_srow, kn_sh, kr_sblk, kr_srow, kr_sh, o_sb, o_sh;
    float sm_scale;
};

template <bool BF16>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    if constexpr (BF16) {
        auto cv = [](float f) -> uint32_t {
            uint32_t x = __float_as_uint(f);
            return (x + 0x7FFFu + ((x >> 16) & 1u)) >> 16;      // RNE; p is finite and >= 0
        };
        return cv(lo) | (cv(hi) << 16);
    } else {
        _Float16 a = (_Float16)lo, b = (_Float16)hi;
        return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
    }
}

template <bool BF16>
__device__ __forceinline__ uint16_t cvt_out(float f)
{
    if constexpr (BF16) {
        uint32_t x = __float_as_uint(f);
        if ((x & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0u;
        return (uint16_t)((x + 0x7FFFu + ((x >> 16) & 1u)) >> 16);
    } else {
        _Float16 a = (_Float16)f;
        return __builtin_bit_cast(uint16_t, a);
    }
}

// issue the LDS-DMA of KV tile `tile` into `buf`; instructions are dealt round-robin to the waves
__device__ __forceinline__ void issue_tile(const MlaParams &p, int b, int kvh, int seq_len, int tile, uint8_t *buf, int wave,
                                           int nwaves, int lane)
{
    const int t0 = tile * kTile;
    const int32_t *bt = p.block_table + (int64_t)b * p.bt_stride;
    for (int i = wave; i < kTile + kTile / 8; i += nwaves) {
        if (i < kTile) {
            int n = t0 + i;
            n = n < seq_len ? n : seq_len - 1;            // rows past the end are masked later; keep the address valid
            const int page = n / p.page_size, row = n - page * p.page_size;
            const uint16_t *src = p.k_nope + (int64_t)bt[page] * p.kn_sblk + (int64_t)row * p.kn_srow + (int64_t)kvh * p.kn_sh;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + lane * 8),
                                             (void __attribute__((address_space(3))) *)(buf + i * kNopeStride), 16, 0, 0);
        } else {
            const int j = i - kTile;                       // 8 keys per instruction
            const int key = j * 8 + (lane >> 3);
            int n = t0 + key;
            n = n < seq_len ? n : seq_len - 1;
            const int page = n / p.page_size, row = n - page * p.page_size;
            const int chunk = (lane & 7) ^ (key & 7);      // XOR swizzle on the source side
            const uint16_t *src = p.k_rope + (int64_t)bt[page] * p.kr_sblk + (int64_t)row * p.kr_srow + (int64_t)kvh * p.kr_sh;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + chunk * 8),
                                             (void __attribute__((address_space(3))) *)(buf + kTile * kNopeStride + j * 8 * kRopeStride),
                                             16, 0, 0);
        }
    }
}

template <bool BF16>
__global__ __launch_bounds__(64 * kMaxWaves) void mla_decode_kernel(MlaParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int split = blockIdx.x;
    const int head_blocks = (p.group + kHeadsPerBlock - 1) / kHeadsPerBlock;
    const int kvh = blockIdx.y / head_blocks, hblk = blockIdx.y % head_blocks;
    const int b = blockIdx.z;
    const int seq_len = p.seq_lens[b];
    const int ntiles = (seq_len + kTile - 1) / kTile;
    const int tps = (ntiles + p.num_splits - 1) / p.num_splits;
    const int t_begin = split * tps;
    const int t_end = min(ntiles, t_begin + tps);
    const int hg = hblk * kHeadsPerBlock + wave * 16 + c16;      // head inside the kv group
    const bool head_ok = hg < p.group;
    const int head = kvh * p.group + hg;

    // Q^T fragments: lane (g, c16) holds q[head][ks*32 + g*8 .. +8]
    s16x8 qf[18];
    {
        const uint16_t *qrow = p.q + (int64_t)b * p.q_sb + (int64_t)head * p.q_sh;
#pragma unroll
        for (int ks = 0; ks < 18; ++ks) {
            if (head_ok) qf[ks] = *(const s16x8 *)(qrow + ks * 32 + g * 8);
            else qf[ks] = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    if (t_begin < t_end) issue_tile(p, b, kvh, seq_len, t_begin, lds, wave, nwaves, lane);
    for (int t = t_begin; t < t_end; ++t) {
        uint8_t *buf = lds + ((t - t_begin) & 1) * kBufBytes;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces of tile t have landed
        __syncthreads();                                      // everyone's have, and tile t-1 is no longer read
        if (t + 1 < t_end) issue_tile(p, b, kvh, seq_len, t + 1, lds + ((t + 1 - t_begin) & 1) * kBufBytes, wave, nwaves, lane);

        // ---- S^T[key, head] = K · Q^T  (4 m-tiles of 16 keys, 18 k-steps of 32 dims)
        f32x4 s[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const s16x8 a = *(const s16x8 *)(buf + (mt * 16 + c16) * kNopeStride + ks * 64 + g * 16);
                s[mt] = mfma16<BF16>(a, qf[ks], s[mt]);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int key = mt * 16 + c16;
                const int chunk = (ks * 4 + g) ^ (key & 7);
                const s16x8 a = *(const s16x8 *)(buf + kTile * kNopeStride + key * kRopeStride + chunk * 16);
                s[mt] = mfma16<BF16>(a, qf[16 + ks], s[mt]);
            }
        }
        // ---- online softmax; lane owns head c16 and keys mt*16 + 4g + r
        const int kbase = t * kTile + 4 * g;
        float tmax = -INFINITY;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = s[mt][r] * p.sm_scale;
                v = (kbase + mt * 16 + r < seq_len) ? v : -INFINITY;
                s[mt][r] = v;
                tmax = fmaxf(tmax, v);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_use);           // m_run = -inf -> 0
        float psum = 0.f;
        uint32_t pk[8];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = __expf(s[mt][r] - m_use);
                psum += e[r];
            }
            pk[mt * 2 + 0] = pack2<BF16>(e[0], e[1]);
            pk[mt * 2 + 1] = pack2<BF16>(e[2], e[3]);
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (alpha != 1.f) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] *= alpha;
        }
        // P^T fragments: k-step kk covers key tiles (2kk, 2kk+1); slots 0..3 / 4..7 of lane group g
        s16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4 w = u32x4{pk[kk * 4 + 0], pk[kk * 4 + 1], pk[kk * 4 + 2], pk[kk * 4 + 3]};
            pf[kk] = __builtin_bit_cast(s16x8, w);
        }
        // ---- O^T[d, head] += V^T · P^T   (V = nope part of the same LDS tile, transposed on the fly)
        const uint8_t *vrow = buf + (4 * g + (c16 >> 2)) * kNopeStride + (c16 & 3) * 8;
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(vrow + (2 * kk) * 16 * kNopeStride + dt * 32));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (s16x4 __attribute__((address_space(3))) *)(vrow + (2 * kk + 1) * 16 * kNopeStride + dt * 32));
                const s16x8 a = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[dt] = mfma16<BF16>(a, pf[kk], acc[dt]);
            }
        }
    }

    // ---- epilogue: lane holds O^T[d = dt*16 + 4g + r][head c16]
    if (!head_ok) return;
    if (p.num_splits == 1) {
        const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
        uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)head * p.o_sh + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) {
            const uint32_t w0 = (uint32_t)cvt_out<BF16>(acc[dt][0] * inv) | ((uint32_t)cvt_out<BF16>(acc[dt][1] * inv) << 16);
            const uint32_t w1 = (uint32_t)cvt_out<BF16>(acc[dt][2] * inv) | ((uint32_t)cvt_out<BF16>(acc[dt][3] * inv) << 16);
            *(uint2 *)(orow + dt * 16) = uint2{w0, w1};
        }
    } else {
        const int64_t idx = ((int64_t)b * p.q_heads + head) * p.num_splits + split;
        float *po = p.ws_o + idx * kDN + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 32; ++dt) *(f32x4 *)(po + dt * 16) = acc[dt];
        if (g == 0) {
            p.ws_ml[idx * 2 + 0] = m_run;
            p.ws_ml[idx * 2 + 1] = l_run;
        }
    }
}

// merge the flash-decoding partials: one wave per (b, head); lane handles 8 of the 512 dims
template <bool BF16>
__global__ __launch_bounds__(256) void mla_merge_kernel(MlaParams p)
{
    const int lane = threadIdx.x & 63;
    const int64_t bh = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (bh >= (int64_t)p.batch * p.q_heads) return;
    const int S = p.num_splits;
    const float *ml = p.ws_ml + bh * S * 2;
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, ml[s * 2]);
    float L = 0.f;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    for (int s = 0; s < S; ++s) {
        const float m = ml[s * 2];
        if (m == -INFINITY) continue;
        const float w = __expf(m - M);
        L += w * ml[s * 2 + 1];
        const float *po = p.ws_o + (bh * S + s) * kDN + lane * 8;
        const f32x4 a = *(const f32x4 *)po, c = *(const f32x4 *)(po + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] += w * a[j];
            o[4 + j] += w * c[j];
        }
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    const int b = (int)(bh / p.q_heads), h = (int)(bh % p.q_heads);
    uint16_t *orow = p.out + (int64_t)b * p.o_sb + (int64_t)h * p.o_sh + lane * 8;
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        w[j] = (uint32_t)cvt_out<BF16>(o[2 * j] * inv) | ((uint32_t)cvt_out<BF16>(o[2 * j + 1] * inv) << 16);
    *(u32x4 *)orow = w;
}

}  // namespace mi_sgl

using namespace mi_sgl;

extern "C" const char *mi_sgl_kernels_version(void) { return "mi_sgl_kernels 0.1 gfx950"; }

extern "C" size_t mi_mla_decode_workspace(int batch, int q_heads, int num_splits)
{
    if (num_splits <= 1) return 0;
    return (size_t)batch * q_heads * num_splits * (kDN + 2) * sizeof(float);
}

extern "C" int mi_mla_decode_num_splits(int batch, int q_heads, int kv_heads, int max_seq_len)
{
    if (batch <= 0 || q_heads <= 0 || kv_heads <= 0 || max_seq_len <= 0) return 1;
    const int group = q_heads / kv_heads;
    const long long wgs = (long long)batch * kv_heads * ((group + kHeadsPerBlock - 1) / kHeadsPerBlock);
    const int ntiles = (max_seq_len + kTile - 1) / kTile;
    int s = (int)((256 + wgs - 1) / wgs);          // at least one workgroup per CU
    const int cap = ntiles / 4 > 1 ? ntiles / 4 : 1;   // keep >= 4 tiles (256 keys) per split
    if (s > cap) s = cap;
    if (s > 64) s = 64;
    return s < 1 ? 1 : s;
}

extern "C" int mi_mla_decode(const void *q, const void *k_nope, const void *k_rope, void *out, const int32_t *kv_seq_lens,
                             const int32_t *block_table, int batch, int q_heads, int kv_heads, int page_size, int bt_stride,
                             int max_seq_len, int64_t q_stride_b, int64_t q_stride_h, int64_t kn_stride_blk,
                             int64_t kn_stride_row, int64_t kn_stride_h, int64_t kr_stride_blk, int64_t kr_stride_row,
                             int64_t kr_stride_h, int64_t o_stride_b, int64_t o_stride_h, float sm_scale, int dtype,
                             int num_splits, void *workspace, size_t workspace_bytes, void *stream)
{
    if (batch < 0 || q_heads <= 0 || kv_heads <= 0 || q_heads % kv_heads || page_size <= 0 || bt_stride <= 0) return MI_SGL_EINVAL;
    if (batch == 0) return MI_SGL_OK;
    if (!q || !k_nope || !k_rope || !out || !kv_seq_lens || !block_table) return MI_SGL_EINVAL;
    if (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) return MI_SGL_EINVAL;
    if ((q_stride_h % 8) || (q_stride_b % 8) || (kn_stride_row % 8) || (kn_stride_blk % 8) || (kn_stride_h % 8) ||
        (kr_stride_row % 8) || (kr_stride_blk % 8) || (kr_stride_h % 8) || (o_stride_h % 8) || (o_stride_b % 8))
        return MI_SGL_EINVAL;      // 16-byte vector accesses
    if (num_splits <= 0) num_splits = mi_mla_decode_num_splits(batch, q_heads, kv_heads, max_seq_len);
    if (num_splits > 1 && (!workspace || workspace_bytes < mi_mla_decode_workspace(batch, q_heads, num_splits))) return MI_SGL_EINVAL;
    MlaParams p;
    p.q = (const uint16_t *)q, p.k_nope = (const uint16_t *)k_nope, p.k_rope = (const uint16_t *)k_rope;
    p.out = (uint16_t *)out, p.seq_lens = kv_seq_lens, p.block_table = block_table;
    p.ws_o = (float *)workspace;
    p.ws_ml = p.ws_o ? p.ws_o + (size_t)batch * q_heads * num_splits * kDN : nullptr;
    p.batch = batch, p.q_heads = q_heads, p.kv_heads = kv_heads, p.group = q_heads / kv_heads, p.page_size = page_size;
    p.bt_stride = bt_stride, p.num_splits = num_splits;
    p.q_sb = q_stride_b, p.q_sh = q_stride_h, p.kn_sblk = kn_stride_blk, p.kn_srow = kn_stride_row, p.kn_sh = kn_stride_h;
    p.kr_sblk = kr_stride_blk, p.kr_srow = kr_stride_row, p.kr_sh = kr_stride_h, p.o_sb = o_stride_b, p.o_sh = o_stride_h;
    p.sm_scale = sm_scale;
    hipStream_t st = (hipStream_t)stream;
    const int head_blocks = (p.group + kHeadsPerBlock - 1) / kHeadsPerBlock;
    const int heads_in_block = p.group < kHeadsPerBlock ? p.group : kHeadsPerBlock;
    const int nwaves = (heads_in_block + 15) / 16;
    dim3 grid(num_splits, kv_heads * head_blocks, batch);
    const size_t lds = 2 * (size_t)kBufBytes;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)mla_decode_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)mla_decode_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    if (dtype == MI_DTYPE_BF16) mla_decode_kernel<true><<<grid, 64 * nwaves, lds, st>>>(p);
    else mla_decode_kernel<false><<<grid, 64 * nwaves, lds, st>>>(p);
    if (num_splits > 1) {
        const long long bh = (long long)batch * q_heads;
        const int blocks = (int)((bh + 3) / 4);
        if (dtype == MI_DTYPE_BF16) mla_merge_kernel<true><<<blocks, 256, 0, st>>>(p);
        else mla_merge_kernel<false><<<blocks, 256, 0, st>>>(p);
    }
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}
