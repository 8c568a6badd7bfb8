// Grouped INT8 expert GEMMs of fused_deep_moe for gfx950 (v_mfma_i32_16x16x64_i8).
// Replaces the "catlass/act" AscendC GEMM templates the reference fuses into one MIX kernel
// (csrc/deepep/ops/op_kernel/fused_deep_moe.h:336-427): GEMM1 int8[R,H] x int8[H,2I] -> i32 with the per-token dequant +
// SwiGLU epilogue (ops/utils/op_kernel/operator/epilogue/block/block_epilogue_per_token_dequant_swiglu.h:250-269),
// the per-row requantisation (.../gemm/kernel/grouped_matmul_slice_m_per_token_dequant_swiglu_quant_multistage_workspace.h:199-265)
// and GEMM2 int8[R,I] x int8[I,H] with per-token x per-channel dequant to bf16.
//
// MI355X design: weights are consumed as [expert][N][K] (K contiguous), activations as [row][K]: both MFMA operands are
// 16-byte K-slices, so tiles go global -> registers -> LDS with 16-B accesses only, XOR-swizzled (chunk ^ (row & 7)) so the
// ds_read_b128 of 16 rows x 128-B stride is conflict-free.  Workgroup tile 128(M) x 128(N) x 128(K bytes), 4 waves stacked in M
// (32 x 128 each: 2 x 8 MFMA tiles = 64 accumulator registers) so a wave owns BOTH the gate (columns 0-63) and up (64-127)
// halves of a fusion tile and the SwiGLU epilogue needs no exchange.  Expert row ranges come from the device-side
// cumulative counts, so the same launch serves low-latency (no host sync) and normal mode; idle tile slots exit at once.
// Bound: MFMA int8 for prefill-size groups (2*M*N*K ops), HBM (weights once: L*N*K bytes) for decode-size groups.
#include "ep_common.h"

namespace mi_ep {

constexpr int BM = 128, BN = 128, BK = 128;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const int8_t *a;          // [M_cap, K]
    const float *a_scale;     // [M_cap]
    const int8_t *w;          // [L, N, K]
    const float *w_scale;     // [L, N]
    const int32_t *cum;       // inclusive cumulative row counts; expert e ends at cum[(e + 1) * cum_stride - 1]
    int cum_stride, L, M_cap, K, N;
    void *out;                // mode 0: float [M_cap, N/2]; mode 1: bf16 [M_cap, N]
};

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 4); }

template <int MODE>
__global__ __launch_bounds__(256) void grouped_gemm_i8_kernel(GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];    // 2 x (A 16 KB + B 16 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
    // which (expert, m-tile) is tile slot blockIdx.y?
    int e = -1, row0 = 0, rows = 0;
    {
        int slot = blockIdx.y, start = 0;
        for (int i = 0; i < p.L; ++i) {
            const int end = p.cum[(i + 1) * p.cum_stride - 1];
            const int cnt = end - start;
            const int tiles = (cnt + BM - 1) / BM;
            if (slot < tiles) {
                e = i;
                row0 = start + slot * BM;
                rows = min(BM, cnt - slot * BM);
                break;
            }
            slot -= tiles;
            start = end;
        }
    }
    if (e < 0) return;
    const int n0 = blockIdx.x * BN;
    const int8_t *wbase = p.w + ((size_t)e * p.N + n0) * p.K;
    const int8_t *abase = p.a + (size_t)row0 * p.K;

    i32x4 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = i32x4{0, 0, 0, 0};

    // staging: thread t moves chunks c = t + 256*i (i < 4) of each tile: row = c / 8, 16-B chunk = c % 8
    u32x4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, r = c >> 3, ch = c & 7;
            ra[i] = (r < rows) ? *(const u32x4 *)(abase + (size_t)r * p.K + k0 + ch * 16) : u32x4{0, 0, 0, 0};
            rb[i] = *(const u32x4 *)(wbase + (size_t)r * p.K + k0 + ch * 16);
        }
    };
    auto lstore = [&](uint8_t *buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i, r = c >> 3, ch = c & 7;
            *(u32x4 *)(buf + swz(r, ch)) = ra[i];
            *(u32x4 *)(buf + BM * BK + swz(r, ch)) = rb[i];
        }
    };
    const int nk = p.K / BK;
    gload(0);
    lstore(lds);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        uint8_t *buf = lds + (kt & 1) * (BM * BK + BN * BK);
        if (kt + 1 < nk) gload((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4 af[2], bf[8];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                af[mt] = *(const i32x4 *)(buf + swz(wave * 32 + mt * 16 + c16, ks * 4 + g));
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) bf[nt] = *(const i32x4 *)(buf + BM * BK + swz(nt * 16 + c16, ks * 4 + g));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 8; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(lds + ((kt + 1) & 1) * (BM * BK + BN * BK));
        __syncthreads();
    }

    // ---- epilogue: lane holds C[row = wave*32 + mt*16 + 4g + r][col = nt*16 + c16]
    const float *ws = p.w_scale + (size_t)e * p.N + n0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lr = wave * 32 + mt * 16 + 4 * g + r;
            if (lr >= rows) continue;
            const size_t grow = (size_t)row0 + lr;
            const float as = p.a_scale[grow];
            if (MODE == 0) {
                // fusion tile: columns 0-63 gate, 64-127 up (weights pre-permuted, reference test_fused_deep_moe.py:75-86)
                float *orow = (float *)p.out + grow * (size_t)(p.N / 2) + blockIdx.x * (BN / 2);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float gate = ((float)acc[mt][nt][r] * ws[nt * 16 + c16]) * as;
                    const float up = ((float)acc[mt][nt + 4][r] * ws[64 + nt * 16 + c16]) * as;
                    orow[nt * 16 + c16] = up * (gate / (1.0f + __expf(-gate)));
                }
            } else {
                uint16_t *orow = (uint16_t *)p.out + grow * (size_t)p.N + n0;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt)
                    orow[nt * 16 + c16] = (uint16_t)f32_to_bf16_rne(((float)acc[mt][nt][r] * ws[nt * 16 + c16]) * as);
            }
        }
}

// per-row symmetric requantisation of the SwiGLU output: q = rint((v * 127) * (1 / rowmax)), scale = rowmax / 127
// (reference ...swiglu_quant_multistage_workspace.h:199-265).  One wave per row.
__global__ __launch_bounds__(256) void rowquant_kernel(const float *__restrict__ v, const int32_t *__restrict__ total_dev, int M_cap,
                                                      int I, int8_t *__restrict__ q, float *__restrict__ scale)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int total = min(*total_dev, M_cap);
    if (row >= total) return;
    const float *vr = v + row * (long long)I;
    float amax = 0.f;
    for (int i = lane * 4; i < I; i += 256) {
        const float4 x = *(const float4 *)(vr + i);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w))));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 1.0f / amax : 0.f;
    if (lane == 0) scale[row] = amax / 127.0f;
    for (int i = lane * 4; i < I; i += 256) {
        const float4 x = *(const float4 *)(vr + i);
        const int a = (int)rintf((x.x * 127.0f) * inv), b = (int)rintf((x.y * 127.0f) * inv);
        const int c = (int)rintf((x.z * 127.0f) * inv), d = (int)rintf((x.w * 127.0f) * inv);
        *(uint32_t *)(q + row * (long long)I + i) =
            (uint32_t)(a & 0xFF) | ((uint32_t)(b & 0xFF) << 8) | ((uint32_t)(c & 0xFF) << 16) | ((uint32_t)(d & 0xFF) << 24);
    }
}

}  // namespace mi_ep

using namespace mi_ep;

static int gemm_launch(int mode, const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale, const int32_t *cum,
                       int cum_stride, int L, int M_cap, int K, int N, void *out, void *stream)
{
    if (!a || !a_scale || !w || !w_scale || !cum || !out || L <= 0 || L > 1024 || M_cap <= 0 || K <= 0 || K % BK || N <= 0 ||
        N % BN || cum_stride <= 0)
        return MI_EP_EINVAL;
    GemmArgs p{a, a_scale, w, w_scale, cum, cum_stride, L, M_cap, K, N, out};
    dim3 grid(N / BN, (M_cap + BM - 1) / BM + L);
    const size_t lds = 2 * (size_t)(BM * BK + BN * BK);
    if (mode == 0) grouped_gemm_i8_kernel<0><<<grid, 256, lds, (hipStream_t)stream>>>(p);
    else grouped_gemm_i8_kernel<1><<<grid, 256, lds, (hipStream_t)stream>>>(p);
    return launch_status();
}

extern "C" int mi_ep_moe_gemm1_swiglu(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale,
                                      const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap, int hidden,
                                      int two_i, float *out, void *stream)
{
    return gemm_launch(0, a, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, hidden, two_i, out, stream);
}

extern "C" int mi_ep_moe_rowquant(const float *v, const int32_t *total_rows_dev, int rows_cap, int inter, int8_t *q, float *scale,
                                  void *stream)
{
    if (!v || !total_rows_dev || !q || !scale || rows_cap <= 0 || inter <= 0 || inter % 4) return MI_EP_EINVAL;
    rowquant_kernel<<<(rows_cap + 3) / 4, 256, 0, (hipStream_t)stream>>>(v, total_rows_dev, rows_cap, inter, q, scale);
    return launch_status();
}

extern "C" int mi_ep_moe_gemm2(const int8_t *a, const float *a_scale, const int8_t *w, const float *w_scale,
                               const int32_t *row_cumsum, int cum_stride, int num_local_experts, int rows_cap, int inter,
                               int hidden, void *out_bf16, void *stream)
{
    return gemm_launch(1, a, a_scale, w, w_scale, row_cumsum, cum_stride, num_local_experts, rows_cap, inter, hidden, out_bf16, stream);
}
