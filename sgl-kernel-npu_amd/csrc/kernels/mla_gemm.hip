// Hand-written skinny GEMMs of torch.ops.npu.mla_preprocess for gfx950 (tokens <= 1024, typically 1..128: the op is bound by
// streaming ~70 MB of weights, not by MFMA).  Reference: the AscendC MIX kernel runs quant -> INT8 GEMM -> RMSNorm -> INT8 GEMM
// -> per-head BMM -> RoPE -> cache write as one op (csrc/mla_preprocess/op_kernel/mla_preprocess_mix_bf16.hpp:285,2762,2814;
// host csrc/mla_preprocess/op_host/mla_preprocess.cpp:623-704); arithmetic per the test golden golden2_pytorch
// (tests/python/sgl_kernel_npu/test_mla_preprocess.py:407-483).
//
//   mi_mla_pre_gemm_i8   C[M, N] (+)= A[M, K] int8 x W[N, K]^T int8      (v_mfma_i32_16x16x64_i8)
//       mode 0 (GEMM1, K = hidden): split-K over 512-byte chunks, one chunk per workgroup, each writing its partial product
//                C[chunk][M][N]; the consumer (pre_mid) adds the chunks -- integer adds commute, so the result is exact;
//       mode 1 (GEMM2, K = 1536):  the whole K in one workgroup, epilogue y = dtype((float(c + bias[n])) * descale[n]).
//   mi_mla_pre_bmm_rope  per head h: q_out0[:, h, :] = Y[:, h, 0:128] x wuk_t[h]^T  (v_mfma_f32_16x16x32_bf16 / f16) and
//                        q_out1[:, h, :] = rope_half(Y[:, h, 128:192]).
//
// MI355X design.  What has to move is the weight matrix, once, at HBM rate, with every CU pulling its share:
//  * a workgroup (4 waves) owns BN weight rows x one K-chunk; each weight row's chunk is 512 contiguous bytes and travels
//    global -> LDS with ONE LDS-DMA instruction (global_load_lds_dwordx4, 32 lanes x 16 B: row-coalesced, no staging
//    registers).  Rows sit 528 B apart in LDS, so the 16 rows x 16 B of a ds_read_b128 operand fetch hit 64 distinct banks.
//    (Streaming the weights straight into MFMA fragments -- 64-byte pieces of 16 rows per load -- was measured earlier at
//    25-43 us per GEMM: the weights have to be staged row-coalesced.)
//  * the activations are tiny (tokens x K int8, L2-resident): every wave keeps the A fragments of its 32 token rows for the
//    current chunk in registers (direct 16-byte loads in MFMA layout) and multiplies them against all BN columns, so an LDS
//    byte of weights is read once per wave and a token block of 128 rows costs one pass over the weights;
//  * grids: GEMM1 = 17 column tiles x 14 K-chunks = 238 workgroups at hidden 7168; GEMM2 = 384 column tiles of 64; the BMM
//    512 (head x column quarter): every launch fills the 256 CUs once.
// Bound: HBM (weights once: 15.1 + 37.7 + 16.8 MB at 128 heads); algorithmic bytes per launch = N*K (+ M*K per column tile
// from L2).
#include "device_once.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi_sgl_kernels.h"
#include "mla_pre_dev.h"

namespace mi_sgl {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kKC = 512;             // bytes of K per weight-row DMA (one LDS-DMA instruction of 32 lanes)
constexpr int kRowStride = kKC + 16; // LDS row stride: consecutive rows start 4 banks apart
constexpr int kBM = 128;             // token rows per workgroup: 4 waves x 2 MFMA row tiles

template <bool BF16>
__device__ __forceinline__ float ldh16(uint16_t bits)
{
    if constexpr (BF16) return __uint_as_float((uint32_t)bits << 16);
    else return (float)__builtin_bit_cast(_Float16, bits);
}
template <bool BF16>
__device__ __forceinline__ uint16_t sth16(float f)
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

// lane l moves 16 B from its own global address to LDS byte `dst` + 16 l (M0 carries the wave-uniform destination)
__device__ __forceinline__ void dma16(uint32_t dst, const void *vaddr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(vaddr) : "memory");
}

// MODE 0: one int32 partial product per K-chunk (split-K, by = chunk);  MODE 1: whole K, dequant epilogue into the I/O dtype.
// NW waves x MT row tiles of 16 token rows = kBM rows per workgroup: (4, 2) in the stand-alone launch -- a weight fragment read from LDS
// feeds two MFMAs --, (8, 1) inside the one-launch form, whose workgroups have eight waves (every wave must reach the barriers).
// SC1: the partial products leave through device-scope (write-through) stores, for a consumer in the same launch.
// PRE: the weight rows of the (first) K-chunk were requested by the caller (skinny_i8_issue_weights) before it did something else.
template <int BN, int NW>
__device__ __forceinline__ void skinny_i8_issue_weights(const int8_t *__restrict__ W, int N, int K, int n0, int k0, int klen, uint32_t lds_base)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll 4
    for (int r = 0; r < BN / NW; ++r) {
        const int row = wave * (BN / NW) + r;
        const int8_t *src = W + (size_t)min(n0 + row, N - 1) * K + k0 + lane * 16;
        if (lane * 16 < klen) dma16(lds_base + (uint32_t)(row * kRowStride), src);
    }
}
template <int MODE, int BN, bool BF16, int NW, int MT, bool SC1, bool PRE = false>
__device__ __forceinline__ void skinny_i8_body(const int8_t *__restrict__ A, int M, int K, const int8_t *__restrict__ W, int N,
                                               int32_t *__restrict__ C, const int32_t *__restrict__ bias, const float *__restrict__ descale,
                                               const float *__restrict__ row_scale, uint16_t *__restrict__ Y, int bx, int by, int bz,
                                               uint8_t *lds /*[BN][kRowStride]*/)
{
    static_assert(NW * MT * 16 == kBM && BN % NW == 0, "row plan");
    constexpr int NT = BN / 16;                       // MFMA column tiles, all of them handled by every wave
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = bx * BN, m0 = bz * kBM + wave * (MT * 16);
    const int chunks = (K + kKC - 1) / kKC;
    const int c_begin = MODE == 0 ? by : 0, c_end = MODE == 0 ? by + 1 : chunks;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);

    i32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = i32x4{0, 0, 0, 0};

    const int8_t *arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = A + (size_t)min(m0 + mt * 16 + c16, M - 1) * K + g * 16;      // rows past M: any valid row

    for (int c = c_begin; c < c_end; ++c) {
        const int k0 = c * kKC;
        const int klen = min(kKC, K - k0);            // multiple of 64
        // weights: wave w moves rows w*BN/NW ..; lane l < klen/16 carries 16 B of the row's chunk
        if (!(PRE && c == c_begin)) skinny_i8_issue_weights<BN, NW>(W, N, K, n0, k0, klen, lds_base);
        // activations of this wave's token rows, straight into MFMA operand layout (L2-resident)
        // Every column tile of a K-chunk reads the SAME activation lines; started in the same order by every workgroup they all
        // queue on one L2 channel at a time.  Workgroup x starts its sweep x k-steps into the chunk (and wraps).
        const int nks = klen / 64;
        const int rot = bx % nks;
        i32x4 af[MT][kKC / 64];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < kKC / 64; ++ks) {
                const int kq = ks + rot < nks ? ks + rot : ks + rot - nks;
                af[mt][ks] = ks < nks ? *(const i32x4 *)(arow[mt] + k0 + kq * 64) : i32x4{0, 0, 0, 0};
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kKC / 64; ++ks) {
            if (ks >= nks) break;                     // wave-uniform
            const int kq = ks + rot < nks ? ks + rot : ks + rot - nks;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const i32x4 bf = *(const i32x4 *)(lds + (nt * 16 + c16) * kRowStride + kq * 64 + g * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[mt][ks], bf, acc[mt][nt], 0, 0, 0);
            }
        }
        __syncthreads();                              // everybody is done with the stage before the next chunk lands
    }

    // lane holds C[row = m0 + mt*16 + 4g + r][col = n0 + nt*16 + c16]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mt * 16 + 4 * g + r;
            if (row >= M) continue;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + nt * 16 + c16;
                if (col >= N) continue;
                if (MODE == 0) {
                    // one partial product per K-chunk: C[chunk][row][col] (plain 64-byte-segment stores; device-scope atomics
                    // would have to leave the XCD-local L2 and ran 3x slower than the whole GEMM)
                    int32_t *dst = C + ((size_t)by * M + row) * N + col;
                    if constexpr (SC1) __hip_atomic_store(dst, acc[mt][nt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else *dst = acc[mt][nt][r];
                } else {
                    float y = (float)(acc[mt][nt][r] + (bias ? bias[col] : 0)) * descale[col];
                    if (row_scale) y = y * row_scale[row];      // per_token_quant_symm: the token's own scale (hpp:2273-2279)
                    Y[(size_t)row * N + col] = sth16<BF16>(y);
                }
            }
        }
}
// (Round 4, measured and dropped: GEMM1 workgroups of 64 columns x 2 chunks or 32 columns x 4 chunks -- the same 64 KB of weights per
//  workgroup, all chunks requested up front, one partial product per GROUP of chunks, so the partials shrink from 15 MB to 7.6 / 4.3 MB and
//  pre_mid sums them in one round trip.  The whole op, queued back to back at 128 tokens: 42.5 us against 43.0 (two chunks), 50 us (four: every
//  workgroup then reads four times the activation rows from L2).  The partial products are not what GEMM1 + pre_mid wait for.)
template <int MODE, int BN, bool BF16>
__global__ __launch_bounds__(256) void skinny_i8_kernel(const int8_t *__restrict__ A, int M, int K, const int8_t *__restrict__ W, int N,
                                                       int32_t *__restrict__ C, const int32_t *__restrict__ bias,
                                                       const float *__restrict__ descale, const float *__restrict__ row_scale,
                                                       uint16_t *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // [BN][kRowStride]
    skinny_i8_body<MODE, BN, BF16, 4, 2, false>(A, M, K, W, N, C, bias, descale, row_scale, Y, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, lds);
}


// GEMM2 (K = 1536): a workgroup owns BN weight rows for the WHOLE K (BN x 1536 B in LDS, rows 1552 B apart: conflict-free like
// above) and all 128 token rows: 4 waves x 2 MFMA row tiles, whose 48 A fragments (the whole K) sit in registers -- every
// weight fragment read from LDS feeds two MFMAs (with one row tile per wave the kernel was LDS-read bound: 12.7 us of its
// 22 us were compute with NO global traffic at all).  All three 512-byte K-chunks (weights by LDS-DMA, activations by direct
// loads) are requested up front in chunk order; chunk c is multiplied as soon as it has landed (vmcnt + one barrier per chunk)
// while the later chunks are still in flight.  With BN = 96 the 24576 output columns of 128 heads are exactly 256 workgroups:
// one per CU, one round.  The bf16 tile leaves through LDS (the weight stage is dead by then) as 16-byte row segments.
constexpr int kK2 = 1536, kRow2 = kK2 + 16;
template <int NT, bool BF16>
__global__ __launch_bounds__(256) void skinny_i8_k1536_kernel(const int8_t *__restrict__ A, int M, const int8_t *__restrict__ W, int N,
                                                             const int32_t *__restrict__ bias, const float *__restrict__ descale,
                                                             const float *__restrict__ row_scale, uint16_t *__restrict__ Y)
{
    constexpr int BN = NT * 16, KS = kKC / 64;       // 8 k-steps per chunk, 3 chunks
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // [BN][kRow2]; reused for the output tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.z * kBM + wave * 32;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);
    const int8_t *arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) arow[mt] = A + (size_t)min(m0 + mt * 16 + c16, M - 1) * kK2 + g * 16;
    // all 256 workgroups read the same activations: each starts its sweep at a different k-step of the chunk (and wraps)
    const int rot = blockIdx.x % KS;
    constexpr int kRowsPerWave = BN / 4;             // DMA instructions per wave and chunk
    i32x4 af[3][2][KS];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int r = 0; r < kRowsPerWave; ++r) {
            const int row = wave * kRowsPerWave + r;
            const int8_t *src = W + (size_t)min(n0 + row, N - 1) * kK2 + c * kKC + lane * 16;
            if (lane < 32) dma16(lds_base + (uint32_t)(row * kRow2 + c * kKC), src);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int kq = ks + rot < KS ? ks + rot : ks + rot - KS;
                af[c][mt][ks] = *(const i32x4 *)(arow[mt] + c * kKC + kq * 64);
            }
    }
    i32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = i32x4{0, 0, 0, 0};
    constexpr int kPerChunk = kRowsPerWave + 2 * KS;  // vector-memory operations a wave issued per chunk
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        // own requests of chunks <= c have landed (they complete in issue order); the barrier extends that to every wave's rows
        if (c == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kPerChunk < 64 ? 2 * kPerChunk : 63) : "memory");
        else if (c == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerChunk < 64 ? kPerChunk : 63) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kq = ks + rot < KS ? ks + rot : ks + rot - KS;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const i32x4 bf = *(const i32x4 *)(lds + (nt * 16 + c16) * kRow2 + c * kKC + kq * 64 + g * 16);
                acc[0][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[c][0][ks], bf, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[c][1][ks], bf, acc[1][nt], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                  // every wave is done reading the weight stage
    // epilogue: dequant into the I/O dtype, parked in the wave's LDS tile [32 rows][BN cols] (+16 B per row), then whole row
    // segments out, 16 B per lane
    constexpr int kTileRow = BN * 2 + 16;
    uint8_t *tile = lds + wave * (32 * kTileRow);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = min(n0 + nt * 16 + c16, N - 1);
        const float ds = descale[col];
        const int32_t bs = bias ? bias[col] : 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float y = (float)(acc[mt][nt][r] + bs) * ds;
                if (row_scale) y = y * row_scale[min(m0 + mt * 16 + 4 * g + r, M - 1)];      // per_token_quant_symm
                *(uint16_t *)(tile + (mt * 16 + 4 * g + r) * kTileRow + (nt * 16 + c16) * 2) = sth16<BF16>(y);
            }
    }
    // a wave reads back only what it wrote itself (its LDS operations complete in order)
    constexpr int kChunks = BN / 8;                   // 16-byte chunks per row
    for (int i = lane; i < 32 * kChunks; i += 64) {
        const int rl = i / kChunks, ch = i - rl * kChunks;
        const int row = m0 + rl, col = n0 + ch * 8;
        if (row >= M || col >= N) continue;
        const uint4 v = *(const uint4 *)(tile + rl * kTileRow + ch * 16);
        uint16_t *dst = Y + (size_t)row * N + col;
        if (col + 8 <= N && (N % 8) == 0) {
            *(uint4 *)dst = v;
        } else {
            const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 8 && col + j < N; ++j) dst[j] = (uint16_t)(wds[j >> 1] >> (16 * (j & 1)));
        }
    }
}

// per head: q_out0[m, h, n] = sum_k Y[m, h*192 + k] * wuk_t[h, n, k] (k < 128, fp32 accumulate, one rounding to the I/O dtype)
// and the rotate-half RoPE of Y[m, h*192 + 128 ..] -> q_out1.  grid (heads, 4 column quarters, token blocks of 128).
template <bool BF16>
__global__ __launch_bounds__(256) void bmm_rope_kernel(const uint16_t *__restrict__ Y, int M, int Hq, const uint16_t *__restrict__ wuk_t,
                                                      const uint16_t *__restrict__ cosv, const uint16_t *__restrict__ sinv,
                                                      uint16_t *__restrict__ out0, uint16_t *__restrict__ out1,
                                                      const uint16_t *__restrict__ q_nope_scale)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[4][32][128 + 8];      // per wave: 32 rows x 128 columns (+16 B: bank spread)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g = lane >> 4;
    const int h = blockIdx.x, quarter = blockIdx.y, m0 = blockIdx.z * kBM + wave * 32;
    const size_t ystride = (size_t)Hq * 192;
    // everything this wave needs is requested before the first MFMA: 8 activation fragments + 32 weight fragments (16 B each)
    s16x8 af[2][4], bf[8][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const uint16_t *row = Y + (size_t)min(m0 + mt * 16 + c16, M - 1) * ystride + (size_t)h * 192 + g * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) af[mt][ks] = *(const s16x8 *)(row + ks * 32);
    }
    const uint16_t *wh = wuk_t + ((size_t)h * 512 + quarter * 128 + c16) * 128 + g * 8;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bf[nt][ks] = *(const s16x8 *)(wh + (size_t)nt * 16 * 128 + ks * 32);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
        f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (BF16) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[0][ks]), __builtin_bit_cast(bf16x8, bf[nt][ks]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[1][ks]), __builtin_bit_cast(bf16x8, bf[nt][ks]), acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[0][ks]), __builtin_bit_cast(f16x8, bf[nt][ks]), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[1][ks]), __builtin_bit_cast(f16x8, bf[nt][ks]), acc1, 0, 0, 0);
            }
        }
        // lane holds D[row 4g + r][col c16] of both row tiles: park it in the wave's LDS tile (rows become contiguous)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tile[wave][4 * g + r][nt * 16 + c16] = sth16<BF16>(acc0[r]);
            tile[wave][16 + 4 * g + r][nt * 16 + c16] = sth16<BF16>(acc1[r]);
        }
    }
    // a wave only reads back its own tile (LDS operations of one wave complete in order): whole 256-byte row segments out,
    // 16 B per lane, 4 rows per wave-store
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rl = it * 4 + (lane >> 4), chunk = lane & 15;
        const int row = m0 + rl;
        if (row >= M) continue;
        const uint4 v = *(const uint4 *)&tile[wave][rl][chunk * 8];
        if (!q_nope_scale) {
            *(uint4 *)(out0 + ((size_t)row * Hq + h) * 512 + quarter * 128 + chunk * 8) = v;
        } else {
            // cache_mode int8_nzcache: q_out0 is int8 = round(clamp(fp16(q * q_nope_scale[h]))) of the value already rounded to the
            // I/O dtype (quant_per_tensor_muls of the golden, tests/python/sgl_kernel_npu/test_mla_preprocess.py:83-90,466-471)
            const float sc = ldh16<BF16>(q_nope_scale[h]);
            const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
            uint32_t pk[2] = {0u, 0u};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float prod = ldh16<BF16>((uint16_t)(wds[j >> 1] >> (16 * (j & 1)))) * sc;
                asm volatile("" : "+v"(prod));          // the fp32 product is rounded before the fp16 conversion (no v_fma_mixlo folding)
                float hq = (float)(_Float16)prod;
                hq = fminf(fmaxf(hq, -128.f), 127.f);
                pk[j >> 2] |= ((uint32_t)(int)rintf(hq) & 0xFFu) << (8 * (j & 3));
            }
            *(uint2 *)((int8_t *)out0 + ((size_t)row * Hq + h) * 512 + quarter * 128 + chunk * 8) = uint2{pk[0], pk[1]};
        }
    }
    // RoPE of the 64 positional columns (lane = column): column quarter q takes rows 8q .. 8q+7 of the wave's 32; the eight rows
    // are independent, so their loads are all in flight together (a serial row loop cost one memory round trip per row)
    uint16_t px[8], pr[8], pc[8], ps[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = min(m0 + quarter * 8 + i, M - 1);
        const uint16_t *pe = Y + (size_t)row * ystride + (size_t)h * 192 + 128;
        px[i] = pe[lane], pr[i] = pe[lane ^ 32], pc[i] = cosv[(size_t)row * 64 + lane], ps[i] = sinv[(size_t)row * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = m0 + quarter * 8 + i;
        if (row >= M) continue;
        const float x = ldh16<BF16>(px[i]);
        const float rot = lane < 32 ? -ldh16<BF16>(pr[i]) : ldh16<BF16>(pr[i]);
        out1[((size_t)row * Hq + h) * 64 + lane] = sth16<BF16>(x * ldh16<BF16>(pc[i]) + rot * ldh16<BF16>(ps[i]));
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// GEMM2 + per-head BMM + RoPE in ONE launch: a workgroup owns one q head (192 GEMM2 columns = 128 nope | 64 rope) and 128 token rows.
//   phase A  y[128, 192] = dequant(A8[128, 1536] x Wuq[h*192 .., :]^T): the weights stream through a 2-deep ring of 256-byte k-chunks
//            (192 rows x 256 B = 48 KB per stage, THREE stages: a CU alone sustains only (bytes in flight) / (memory latency), and with
//            one chunk in flight this kernel ran 34.7 us; whole-wave LDS-DMA instructions of 4 rows; chunk position XOR row & 15 instead of
//            row padding), the activations sit in registers for the whole K, y goes to an LDS tile in the I/O dtype -- the rounding
//            point of the golden (test_mla_preprocess.py:95-107) -- and never to global memory;
//   phase B  q_out0[:, h, :] = y[:, 0:128] x wuk_t[h]^T with wuk_t[h] streamed through the same ring in four 128-column quarters,
//            q_out1[:, h, :] = rope_half(y[:, 128:192]).
// Same MFMA shapes and accumulation order as skinny_i8_k1536_kernel + bmm_rope_kernel, so the outputs are bit-identical to the
// two-launch path.  (128-row form:) 128 heads = 128 workgroups: half the CUs, each pulling 426 KB, which is what a CU's DMA stream sustains when
// the other half is idle; per launch the 6.3 MB y round trip and one launch latency go away.
// Measured at 128 tokens x 128 heads: 26.6 us (28.8 before the per-head rotation of the piece order) against 22.4 + 15.5 us for
// the two launches.  Shader-clock accounting (s_memtime, one
// workgroup): phase A 28-31k cycles for 6 chunks = 4.7k per 48 KB chunk (1.5k of MFMA issue per SIMD), dequant + RoPE 8k, phase B 15k.
// None of the following moved phase A: two or three stages in the ring, 48 KB chunks made contiguous in memory, one or two barriers per
// chunk, DMA issue staggered between the two waves of a SIMD, a deeper ds_read pipeline.  Ablations (whole kernel, 26.6 us): without
// the weight DMA 28.3 us, without the LDS operand reads 24.4, without the activation loads 24.4 -- it is not waiting for memory (a
// plain stream over 128 CUs reaches 5.6 TB/s, tools/probes/stream_probe.hip); what is left is eight waves meeting at a barrier every
// 48 MFMAs.  4 waves x 32 rows (every weight fragment feeding two MFMAs, one wave per SIMD) ran 48 us.
// Workgroup spans (-DF_SPAN, tools/probes/time_mla_pre_tail.py): the 256 workgroups start within 0.5-1.0 us of each other and run 15.2-16.0 us
// each; the kernel takes 21.5 us between the command processor's timestamps -- ~5 us of every launch here is ramp and drain, not work.
// Round 3: 64-row workgroups (HALF below: 256 workgroups at 128 tokens x 128 heads) 26.6 -> 22.9 us; stamps of one wave (-DF_TIMING,
// tools/probes/time_mla_pre_tail.py; shader clocks, 64-row / 128-row form): first chunk landed for every wave 7.7k / 7.2k -- all CUs ask
// for their first 96 KB at once, 24 MB at HBM rate --, the other five chunks 13.0k / 14.6k, dequant + y tile + RoPE 4.9k / 6.0k,
// phase B 9.0k / 17.2k; total 36k / 46k cycles.  Phase A does not depend on the MFMA or LDS volume per workgroup (halved by HALF, same
// 22k cycles): after the cold start it runs at one chunk per ~2.6k cycles against 0.4k / 0.8k of MFMA issue per SIMD.
#ifdef F_SPAN        // start / end of every workgroup on the 100 MHz clock (launch skew and spread of the workgroup durations)
__device__ unsigned long long g_f_span[1024][2];
#define F_SPAN_AT(i) if (tid == 0 && blockIdx.z == 0 && blockIdx.x < 1024) g_f_span[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime();
#else
#define F_SPAN_AT(i)
#endif
#ifdef F_TIMING
__device__ unsigned long long g_f_dbg[4][64];
#define F_STAMP(i) if (tid == 0 && blockIdx.x < 4 && blockIdx.z == 0) g_f_dbg[blockIdx.x][i] = __builtin_amdgcn_s_memtime();
#else
#define F_STAMP(i)
#endif
constexpr int kF_Chunk = 256, kF_NChunks = kK2 / kF_Chunk, kF_Rows = 192, kF_Stage = kF_Rows * kF_Chunk;      // 48 KB
constexpr int kF_YRow = 192 * 2;                                       // y tile row (bytes); the 16 nope chunks of a row are XOR-swizzled
constexpr int kF_OutRow = 128 * 2 + 16;
constexpr int kF_Slots = 3;                                            // ring depth: two k-chunks (96 KB) in flight per CU
constexpr int kF_Lds = kF_Slots * kF_Stage;                            // 147456; the y tile (128 x 384 B) takes over slot 0 after phase A
static_assert(kBM * kF_YRow <= kF_Stage, "y tile fits a ring slot");
// byte offset of element (row, col) of the y tile: 16-byte chunk col / 8 of the nope part sits at position chunk ^ ((row >> 1) & 7), so
// the 16 rows x 16 B of an MFMA operand fetch cover all 64 banks with 384-byte rows (positional columns 128.. are not swizzled)
__device__ __forceinline__ int ytile_off(int row, int col)
{
    const int chunk = col >> 3;
    const int pos = chunk < 16 ? (chunk ^ ((row >> 1) & 7)) : chunk;
    return row * kF_YRow + pos * 16 + (col & 7) * 2;
}

// HALF: the workgroup takes 64 token rows instead of 128 -- wave w = row tile w & 3, column half w >> 2 (6 of the 12 GEMM2 column tiles,
// 2 of the 4 column tiles of every wuk_t eighth).  Same weight stream per workgroup, half the MFMAs and half the LDS operand reads:
// 128 tokens x 128 heads become 256 workgroups (every CU) instead of 128, and batches of <= 64 tokens stop multiplying padding rows.
// Accumulation order per output element is unchanged, so both forms are bit-identical.
// (body: `h` = head, `bz` = token block; `lds` = the kF_Lds-byte ring.  PRE0: chunk 0 of the head's GEMM2 weights is already in ring
//  slot 0 -- requested by the one-launch form in front of its earlier stages, tail_prefetch0 below.)
template <bool BF16, bool HALF, bool PRE0>
__device__ __forceinline__ void gemm2_bmm_rope_body(const int8_t *__restrict__ A, int M, const int8_t *__restrict__ W, int Hq,
                                                    const int32_t *__restrict__ bias, const float *__restrict__ descale,
                                                    const float *__restrict__ row_scale, const uint16_t *__restrict__ wuk_t,
                                                    const uint16_t *__restrict__ cosv, const uint16_t *__restrict__ sinv,
                                                    uint16_t *__restrict__ out0, uint16_t *__restrict__ out1,
                                                    const uint16_t *__restrict__ q_nope_scale, const int h, const int bz, uint8_t *lds)
{
    // 8 waves, one 16-row MFMA tile each (two waves per SIMD): a wave alone on its SIMD waits out every ds_read_b128 in front of the
    // two MFMAs it feeds -- the first version of this kernel, 4 waves x 32 rows with all 256 VGPRs taken by the activations, ran
    // 48 us, slower than the two launches it replaces
    // `lds`: ring [3][192][256]; slot 0 later: y tile, then output tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    constexpr int NT = HALF ? 6 : 12;                       // GEMM2 column tiles of this wave
    const int rt = HALF ? (wave & 3) : wave, ch = HALF ? (wave >> 2) : 0;      // row tile inside the workgroup, column half
    const int nt0 = ch * NT;
    const int m0 = bz * (HALF ? 64 : kBM) + rt * 16;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);
    uint8_t *ytile = lds + kF_Stage;                       // slot 1
    const int8_t *wh = W + (size_t)h * kF_Rows * kK2;
    const uint16_t *uk = wuk_t + (size_t)h * 512 * 128;

    // ---- DMA plans.  One instruction = 4 rows x 256 B; lane l: row 4 i + l / 16, 16-byte position l % 16, which receives source chunk
    // position ^ (row & 15): a ds_read_b128 of 16 consecutive rows at one k-chunk then touches 16 different bank quads.
    const int drow = lane >> 4, dpos = lane & 15;
    const int rot48 = (h * 11) % 48;
    // (walking the k-chunks in a per-head rotated order on top of that changed nothing: 26.6 us either way)
    auto issue_w = [&](int c, int slot) {                                  // GEMM2 weights, k-chunk c: 48 instructions, 6 per wave
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            // the 48 pieces of a chunk are requested in an order rotated by head: every workgroup's weight block starts 288 KB
            // after its neighbour's, so in lockstep they would all be asking the same few memory channels for the same rows
            int piece = wave * 6 + i + rot48;
            piece = piece >= 48 ? piece - 48 : piece;
            const int row = 4 * piece + drow;
            dma16(lds_base + (uint32_t)(slot * kF_Stage + 4 * piece * kF_Chunk),
                  wh + (size_t)row * kK2 + c * kF_Chunk + ((dpos ^ (row & 15)) << 4));
        }
    };
    // wuk_t[h] travels in eighths: 64 output columns x 256 B = 16 KB = 16 instructions, two per wave; `dst` = LDS byte offset
    auto issue_uk = [&](int e, int dst) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 4 * (wave * 2 + i) + drow;
            dma16(lds_base + (uint32_t)(dst + 4 * (wave * 2 + i) * kF_Chunk), uk + ((size_t)(e * 64 + row) * 128) + ((dpos ^ (row & 15)) << 3));
        }
    };
    constexpr int kEighth = 64 * kF_Chunk;                 // 16 KB
    // where eighth e waits in LDS: 0..2 in slot 0, 3..5 in slot 2, 6 and 7 take the places of 0 and 1 once those have been multiplied
    auto eighth_at = [&](int e) { return e < 3 ? e * kEighth : e < 6 ? 2 * kF_Stage + (e - 3) * kEighth : (e - 6) * kEighth; };
    // dequant operands of this lane's 12 columns and 4 rows: requested before anything else (behind the DMA stream their loads would
    // wait for every piece in flight)
    float dsc[NT], rsc[4];
    int32_t bsv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        dsc[nt] = descale[h * kF_Rows + (nt0 + nt) * 16 + c16];
        bsv[nt] = bias ? bias[h * kF_Rows + (nt0 + nt) * 16 + c16] : 0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // (one-launch form: the scales were written in this launch by other workgroups, several to a cache line: device-scope loads)
        if constexpr (PRE0)
            rsc[r] = row_scale ? __uint_as_float(__hip_atomic_load((const uint32_t *)row_scale + min(m0 + 4 * g + r, M - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 1.f;
        else rsc[r] = row_scale ? row_scale[min(m0 + 4 * g + r, M - 1)] : 1.f;
    }
    // RoPE operands of this lane (row = lane / 4 of the wave's 16, 16 of the 64 positional columns): requested first, used last
    const int rrow = lane >> 2, rq = lane & 3;
    const bool rvalid = m0 + rrow < M;
    uint4 rc[2], rs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        rc[j] = rvalid ? *(const uint4 *)(cosv + (size_t)(m0 + rrow) * 64 + rq * 16 + j * 8) : uint4{0, 0, 0, 0};
        rs[j] = rvalid ? *(const uint4 *)(sinv + (size_t)(m0 + rrow) * 64 + rq * 16 + j * 8) : uint4{0, 0, 0, 0};
    }
    // ---- phase A.  Activations travel WITH the weight chunks: the 4 fragments of chunk c are requested right behind chunk c's DMA
    // pieces (inline asm, so that the explicit vmcnt below is the only wait they get) into a double buffer.  Holding all 24 fragments
    // for the whole K left the compiler two registers' worth of ds_read look-ahead in front of every pair of MFMAs (phase A: 31k
    // cycles against 1.5k of MFMA issue per chunk and SIMD).
    const int8_t *arow = A + (size_t)min(m0 + c16, M - 1) * kK2 + g * 16;
    i32x4 afb[2][4];
    auto issue_a = [&](int c, i32x4 (&dst)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[ks]) : "v"(arow + c * kF_Chunk + ks * 64) : "memory");
    };
    F_STAMP(0)
    F_SPAN_AT(0)
    if constexpr (!PRE0) issue_w(0, 0);
    issue_a(0, afb[0]);
    issue_w(1, 1);
    issue_a(1, afb[1]);
    // the compiler's own loads above (dequant operands, cos / sin) are consumed here in its eyes: its wait lands at the start of the
    // kernel, where it costs nothing extra, instead of in front of the epilogue behind a ring full of requests
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(dsc[nt]), "v"(bsv[nt]));
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(rsc[r]));
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(rc[j].x), "v"(rc[j].y), "v"(rc[j].z), "v"(rc[j].w), "v"(rs[j].x), "v"(rs[j].y), "v"(rs[j].z), "v"(rs[j].w));
    i32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = i32x4{0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < kF_NChunks; ++c) {
        // chunk c and its activation fragments landed: requests complete in issue order, and the one request group younger than chunk
        // c is chunk c + 1 (6 pieces + 4 fragments; behind the last chunk: the 6 pieces of wuk_t's first three eighths).  The
        // fragment registers are operands of the wait so that no MFMA is scheduled above it.
        i32x4(&af)[4] = afb[c & 1];
        if (c + 1 < kF_NChunks) asm volatile("s_waitcnt vmcnt(10)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
        else asm volatile("s_waitcnt vmcnt(6)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
        F_STAMP(1 + 3 * c)
        __syncthreads();                                  // chunk c complete for every wave; the slot of chunk c - 1 is free
        F_STAMP(2 + 3 * c)
        // ONE barrier per chunk: the refill of the freed slot is requested here, by half of the waves before and by the other half
        // after their MFMAs (waves w and w + 4 share a SIMD: one issues DMA while the other multiplies)
        auto refill = [&]() {
            if (c + 2 < kF_NChunks) issue_w(c + 2, (c + 2) % kF_Slots);
            else if (c == 4) {                            // slot 0 (chunk 3) -> wuk_t[h] eighths 0..2; slot 1 (chunk 4) becomes the y tile
                issue_uk(0, eighth_at(0));
                issue_uk(1, eighth_at(1));
                issue_uk(2, eighth_at(2));
            }
        };
        if (wave < 4) refill();
        const uint8_t *st = lds + (c % kF_Slots) * kF_Stage;
        // (a fenced explicit pipeline -- the 12 fragments of k-step ks + 1 requested before the 12 MFMAs of ks -- measured the same
        // wall time: phase A is not waiting for LDS reads)
        {
            // weight fragments kF_Ahead MFMAs ahead of their use (ring of kF_Ahead + 2 registers sets): the compiler's own order was
            // `2 ds_read -> wait -> MFMA -> wait -> MFMA`, one exposed LDS round trip per pair with only two waves on the SIMD
            constexpr int kF_Ahead = 6, kRingF = 8, NF = 4 * NT;
            auto ldf = [&](int i) {
                const int ks = i / NT, nt = nt0 + i % NT;
                return *(const i32x4 *)(st + (nt * 16 + c16) * kF_Chunk + (((4 * ks + g) ^ c16) << 4));
            };
            i32x4 fr[kRingF];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kF_Ahead; ++i) fr[i] = ldf(i);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                if (i + kF_Ahead < NF) fr[(i + kF_Ahead) % kRingF] = ldf(i + kF_Ahead);
                __builtin_amdgcn_sched_barrier(0);
                acc[i % NT] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i / NT], fr[i % kRingF], acc[i % NT], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (wave >= 4) refill();
        if (c + 2 < kF_NChunks) issue_a(c + 2, afb[c & 1]);      // into the buffer this chunk's MFMAs have just read
        F_STAMP(3 + 3 * c)
    }
    __syncthreads();                                      // every wave is done with the last chunk (slot 2)
    issue_uk(3, eighth_at(3));
    issue_uk(4, eighth_at(4));
    issue_uk(5, eighth_at(5));
    // dequant into the I/O dtype -> y tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y = (float)(acc[nt][r] + bsv[nt]) * dsc[nt];
            if (row_scale) y = y * rsc[r];                  // per_token_quant_symm
            *(uint16_t *)(ytile + ytile_off(rt * 16 + 4 * g + r, (nt0 + nt) * 16 + c16)) = sth16<BF16>(y);
        }
    asm volatile("" ::: "memory");                       // 2-byte stores above, 16-byte loads below: keep their order (see moe_gemm.hip)
    if constexpr (HALF) __syncthreads();                  // a row tile of y is written by two waves
    // ---- phase B operands: this wave's 16 rows of y_nope as MFMA A fragments; RoPE of its 16 x 64 positional values
    s16x8 yf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) yf[ks] = *(const s16x8 *)(ytile + ytile_off(rt * 16 + c16, ks * 32 + g * 8));
    if (ch == 0) {   // rotate-half RoPE: lane = (row, 16-column quarter); the partner columns (c ^ 32) are quarter ^ 2 of the same y tile row
        const uint16_t *yrow = (const uint16_t *)(ytile + (rt * 16 + rrow) * kF_YRow) + 128;
        uint4 xv[2], pv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            xv[j] = *(const uint4 *)(yrow + rq * 16 + j * 8);
            pv[j] = *(const uint4 *)(yrow + (rq ^ 2) * 16 + j * 8);
        }
        if (rvalid) {
            uint16_t *dst = out1 + ((size_t)(m0 + rrow) * Hq + h) * 64 + rq * 16;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t xw[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w}, pw[4] = {pv[j].x, pv[j].y, pv[j].z, pv[j].w};
                const uint32_t cw[4] = {rc[j].x, rc[j].y, rc[j].z, rc[j].w}, sw[4] = {rs[j].x, rs[j].y, rs[j].z, rs[j].w};
                uint32_t ow[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    uint32_t o2 = 0;
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) {
                        const float x = ldh16<BF16>((uint16_t)(xw[e] >> (16 * b2)));
                        const float pr = ldh16<BF16>((uint16_t)(pw[e] >> (16 * b2)));
                        const float rot = rq < 2 ? -pr : pr;
                        const float cc = ldh16<BF16>((uint16_t)(cw[e] >> (16 * b2))), ss = ldh16<BF16>((uint16_t)(sw[e] >> (16 * b2)));
                        o2 |= (uint32_t)sth16<BF16>(x * cc + rot * ss) << (16 * b2);
                    }
                    ow[e] = o2;
                }
                *(uint4 *)(dst + j * 8) = uint4{ow[0], ow[1], ow[2], ow[3]};
            }
        }
    }
    F_STAMP(20)
    const float qsc = q_nope_scale ? ldh16<BF16>(q_nope_scale[h]) : 0.f;
    uint8_t *otile = ytile + wave * (16 * kF_OutRow);      // the y tile becomes the waves' output tiles after the barrier of quarter 0
    auto multiply_eighth = [&](int e) {                    // 16 rows x 64 output columns of q_out0 (HALF: this wave's 32 of them)
        const uint8_t *st = lds + eighth_at(e);
        constexpr int NB = HALF ? 2 : 4;                   // column tiles of the eighth this wave multiplies
        const int nb0 = ch * NB;
        // independent accumulation chains (k ascending inside each, as in bmm_rope_kernel): the MFMAs issue back to back
        f32x4 oacc[NB];
#pragma unroll
        for (int nt = 0; nt < NB; ++nt) oacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < NB; ++nt) {
                const s16x8 bf = *(const s16x8 *)(st + ((nb0 + nt) * 16 + c16) * kF_Chunk + (((4 * ks + g) ^ c16) << 4));
                if constexpr (BF16)
                    oacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, yf[ks]), __builtin_bit_cast(bf16x8, bf), oacc[nt], 0, 0, 0);
                else
                    oacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, yf[ks]), __builtin_bit_cast(f16x8, bf), oacc[nt], 0, 0, 0);
            }
#pragma unroll
        for (int nt = 0; nt < NB; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) *(uint16_t *)(otile + (4 * g + r) * kF_OutRow + (nt * 16 + c16) * 2) = sth16<BF16>(oacc[nt][r]);
        asm volatile("" ::: "memory");
        constexpr int kChunks = NB * 2;                    // 16-byte chunks per output tile row
#pragma unroll
        for (int it = 0; it < NB / 2; ++it) {
            const int rl = it * (64 / kChunks) + lane / kChunks, chunk = lane % kChunks;
            const int row = m0 + rl;
            const uint4 v = *(const uint4 *)(otile + rl * kF_OutRow + chunk * 16);
            if (row >= M) continue;
            const int col = e * 64 + nb0 * 16 + chunk * 8;
            if (!q_nope_scale) {
                *(uint4 *)(out0 + ((size_t)row * Hq + h) * 512 + col) = v;
            } else {                                        // int8_nzcache: see bmm_rope_kernel
                const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
                uint32_t pk[2] = {0u, 0u};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float prod = ldh16<BF16>((uint16_t)(wds[j >> 1] >> (16 * (j & 1)))) * qsc;
                    asm volatile("" : "+v"(prod));
                    float hq = (float)(_Float16)prod;
                    hq = fminf(fmaxf(hq, -128.f), 127.f);
                    pk[j >> 2] |= ((uint32_t)(int)rintf(hq) & 0xFFu) << (8 * (j & 3));
                }
                *(uint2 *)((int8_t *)out0 + ((size_t)row * Hq + h) * 512 + col) = uint2{pk[0], pk[1]};
            }
        }
        asm volatile("" ::: "memory");
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // eighths 0..5 landed
    __syncthreads();                                        // ... for every wave; the y tile reads are done, it becomes the output tiles
    multiply_eighth(0);
    multiply_eighth(1);
    __syncthreads();                                        // the places of eighths 0 and 1 are free
    issue_uk(6, eighth_at(6));
    issue_uk(7, eighth_at(7));
#pragma unroll 1
    for (int e = 2; e < 6; ++e) multiply_eighth(e);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    multiply_eighth(6);
    multiply_eighth(7);
    F_STAMP(21)
    F_SPAN_AT(1)
}
template <bool BF16, bool HALF>
__global__ __launch_bounds__(512) void gemm2_bmm_rope_kernel(const int8_t *__restrict__ A, int M, const int8_t *__restrict__ W, int Hq,
                                                            const int32_t *__restrict__ bias, const float *__restrict__ descale,
                                                            const float *__restrict__ row_scale, const uint16_t *__restrict__ wuk_t,
                                                            const uint16_t *__restrict__ cosv, const uint16_t *__restrict__ sinv,
                                                            uint16_t *__restrict__ out0, uint16_t *__restrict__ out1,
                                                            const uint16_t *__restrict__ q_nope_scale)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    gemm2_bmm_rope_body<BF16, HALF, false>(A, M, W, Hq, bias, descale, row_scale, wuk_t, cosv, sinv, out0, out1, q_nope_scale, (int)blockIdx.x,
                                           (int)blockIdx.z, lds);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The whole op in ONE launch (round 4; decode sizes: as many workgroups as the per-head stage has, all resident at once) -- BUILT,
// BIT-IDENTICAL, NOT FASTER, therefore opt-in (MI_MLA_PRE_ONE_LAUNCH=1).
// The four launches are four single-round kernels around three grid-wide dependencies (GEMM1 needs every quantised row; the middle
// stage needs every K-chunk of a token; GEMM2 needs the whole normalised row).  Each boundary costs a launch (2.4 us on a stream,
// profiles/r03_launch_floor.txt) plus a cold start.  Here the same stage bodies run back to back inside one grid, separated by grid
// barriers; every workgroup requests the first 48 KB of its head's GEMM2 weights and the weight rows of its GEMM1 tile before anything
// else; hand-off data between the stages (quantised rows, split-K partial products, requantised q, per-token scales) leaves through
// device-scope write-through stores and is read once, by workgroups that have not touched those lines in this launch: no cache fence.
// Stamps of all 256 workgroups at 128 tokens x 128 heads (100 MHz clock, us since the first workgroup's start; -DMEGA_TIMING):
//   stage 0 done 3.2 | barrier passed 5.0-5.7 | GEMM1 done 13.6 | barrier 16.2-16.9 | middle done 23.8 (median 19.8: one token per
//   workgroup, half of the workgroups idle) | barrier 25.9-26.7 | per-head stage done 42.3  -> 43 us on the device, the four launches'
//   kernels sum to 43.6 us: a barrier costs what it replaces (1.8-2.6 us = the drain of the write-through stores + two flag hops against
//   a 2.4 us launch), and what the stages themselves take is memory-latency chains (two dependent round trips each), not launch overhead.
//   With the GEMM1 weights requested ahead of stage 0 GEMM1 shrinks 8.0 -> 6.1 us and stage 0 grows 2.0 -> 4.0 us (its loads queue behind the
//   DMA).  Between events with a synchronisation per call: 53.0 us against 53.4 us for the four launches (same box).
// All stage bodies are the stand-alone launches' (same arithmetic, same summation orders): the outputs are bit-identical, which the tests assert.
struct MegaArgs {
    // stage 0: quantisation of the hidden states (per tensor: scale0 / zp0; per token: tok0 out)
    const uint16_t *x; int tokens, hidden; const uint16_t *qscale0; const int8_t *qoff0; int8_t *a8; float *tok0;
    // stage 1: GEMM1 split-K
    const int8_t *wdqkv; int32_t *c1; int parts;
    // stage 2: middle
    const int32_t *bias0; const float *descale0; const uint16_t *gamma1, *beta1, *gamma2, *cosv, *sinv; const int32_t *slotmapping;
    const uint16_t *qscale1; const int8_t *qoff1; float eps; int8_t *q8; uint16_t *kv_cache, *kv_cache_rope; float *tok1; int cache_mode, block_size;
    const uint16_t *ctkv_scale;
    // stage 3: GEMM2 + BMM + RoPE per head
    const int8_t *wuq; int q_heads; const int32_t *bias1; const float *descale1; const uint16_t *wuk_t; uint16_t *out0, *out1; const uint16_t *q_nope_scale;
    // grid barriers: one flag word per workgroup and barrier + one "go" word per barrier, tagged with the call's epoch (never cleared)
    uint32_t *flags; uint32_t epoch;
    int per_token;
};
constexpr int kMegaMaxGrid = 1024, kMegaBarriers = 3;
// The call's epoch is DEVICE-resident (a graph replay must not carry a host-side counter): every workgroup reads the count of completed
// calls when it starts -- before it can arrive at the first barrier --, the gatherer bumps it after everybody has arrived at the last.
__device__ __forceinline__ uint32_t mega_epoch(uint32_t *flags, uint32_t *lds_word)
{
    if (threadIdx.x == 0) *lds_word = __hip_atomic_load(flags + kMegaBarriers * (kMegaMaxGrid + 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    __syncthreads();
    return *(volatile uint32_t *)lds_word;
}
__device__ __forceinline__ void mega_barrier(const MegaArgs &p, int which)
{
    const int G = (int)(gridDim.x * gridDim.z), wg = (int)(blockIdx.x + gridDim.x * blockIdx.z);
    uint32_t *mine = p.flags + which * (kMegaMaxGrid + 16), *go = mine + kMegaMaxGrid;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through stores have left the CU
    __syncthreads();
    if (threadIdx.x < 64) {
        // Two hops: every workgroup raises its own flag, the first workgroup's first wave gathers them (lane l: flags l, l + 64, ...) and
        // raises ONE word that the others watch; relaxed device-scope accesses only.  Measured (MEGA_TIMING, 256 workgroups): 1.8-2.6 us
        // from the last workgroup's end of stage to the first workgroup past the barrier, the drain of the write-through stores
        // included.  (One hop -- every workgroup watching all 256 flags -- is slower: 3.2-4.9 us, the polls load the memory system.)
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) __hip_atomic_store(mine + wg, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wg == 0) {
            for (int i = (int)threadIdx.x; i < G; i += 64)
                while (__hip_atomic_load(mine + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) __builtin_trap();      // 2 s: a workgroup never became resident
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every lane has seen its flags (the wave reconverges behind the loops)
            if (threadIdx.x == 0) {
                if (which == kMegaBarriers - 1)                 // everybody read the call counter long ago: the next call's epoch is one higher
                    __hip_atomic_store(p.flags + kMegaBarriers * (kMegaMaxGrid + 16), p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(go, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (threadIdx.x == 0) {
            while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) __builtin_trap();
            }
        }
    }
    __syncthreads();
}
#ifdef MEGA_TIMING       // 100 MHz stamps per workgroup: start, after every stage and barrier (tools/probes/time_mla_pre_mega.py)
__device__ unsigned long long g_mega_t[1024][8];
#define MEGA_T(i) if (threadIdx.x == 0) g_mega_t[blockIdx.x + gridDim.x * blockIdx.z][i] = __builtin_amdgcn_s_memrealtime();
#else
#define MEGA_T(i)
#endif
template <bool BF16, bool HALF>
__global__ __launch_bounds__(512) void mla_pre_one_launch_kernel(MegaArgs p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // kF_Lds bytes: the per-head stage's ring
    const int G = (int)(gridDim.x * gridDim.z), wg = (int)(blockIdx.x + gridDim.x * blockIdx.z);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);
    MEGA_T(0)
    // (per-tensor quantisation: this thread's eight hidden-state values are requested in front of the weight DMAs below -- vector-memory
    //  results are waited for in issue order, so a load behind 22 DMA pieces would wait for all of them)
    const long long n8 = (long long)p.tokens * p.hidden, i8 = ((long long)wg * 512 + tid) * 8;
    uint4 xv0 = uint4{0, 0, 0, 0};
    if (!p.per_token && i8 < n8) xv0 = *(const uint4 *)(p.x + i8);
    // first of all: k-chunk 0 of this head's GEMM2 weights into ring slot 0 (the piece order of the per-head stage's issue_w(0, 0));
    // the earlier stages keep their scratch behind it
    {
        const int h = (int)blockIdx.x, drow = lane >> 4, dpos = lane & 15, rot48 = (h * 11) % 48;
        const int8_t *wh = p.wuq + (size_t)h * kF_Rows * kK2;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int piece = wave * 6 + i + rot48;
            piece = piece >= 48 ? piece - 48 : piece;
            const int row = 4 * piece + drow;
            dma16(lds_base + (uint32_t)(4 * piece * kF_Chunk), wh + (size_t)row * kK2 + ((dpos ^ (row & 15)) << 4));
        }
    }
    uint8_t *scratch = lds + kF_Stage;                          // 96 KB behind slot 0 for the earlier stages
    // ... and the weight rows of this workgroup's (first) GEMM1 tile: they do not depend on the quantised activations either
    constexpr int kBN1 = 128;
    const int gx1 = (2112 + kBN1 - 1) / kBN1, gy1 = p.parts, gz1 = (p.tokens + kBM - 1) / kBM;
    const bool has_v = wg < gx1 * gy1 * gz1;
    if (has_v) {
        const int k0 = ((wg / gx1) % gy1) * kKC;
        skinny_i8_issue_weights<kBN1, 8>(p.wdqkv, 2112, p.hidden, (wg % gx1) * kBN1, k0, min(kKC, p.hidden - k0), lds_base + (uint32_t)kF_Stage);
    }
    p.epoch = mega_epoch(p.flags, (uint32_t *)(lds + kF_Lds - 16));
    // ---- stage 0: quantise the hidden states
    if (p.per_token) {
        for (int n = wg; n < p.tokens; n += G) pre_quant_token_body<BF16, 512, true>(p.x, p.hidden, p.a8, p.tok0, n, (float *)(lds + kF_Lds - 64));
    } else {
        const float scale = ldh<BF16>(p.qscale0[0]), zp = (float)p.qoff0[0];
        if (i8 < n8) pre_quant_eight_loaded<BF16, true>(xv0, scale, zp, i8, p.a8);
        for (long long i = i8 + (long long)G * 512 * 8; i < n8; i += (long long)G * 512 * 8) pre_quant_eight<BF16, true>(p.x, scale, zp, i, p.a8);
    }
    MEGA_T(1)
    mega_barrier(p, 0);
    MEGA_T(2)
    // ---- stage 1: GEMM1, split-K: virtual workgroup v = (column tile, K-chunk, token block)
    {
        if (has_v)
            skinny_i8_body<0, kBN1, BF16, 8, 1, true, true>(p.a8, p.tokens, p.hidden, p.wdqkv, 2112, p.c1, nullptr, nullptr, nullptr, nullptr, wg % gx1,
                                                            (wg / gx1) % gy1, wg / (gx1 * gy1), scratch);
        for (int v = wg + G; v < gx1 * gy1 * gz1; v += G)
            skinny_i8_body<0, kBN1, BF16, 8, 1, true>(p.a8, p.tokens, p.hidden, p.wdqkv, 2112, p.c1, nullptr, nullptr, nullptr, nullptr, v % gx1,
                                                      (v / gx1) % gy1, v / (gx1 * gy1), scratch);
    }
    MEGA_T(3)
    mega_barrier(p, 1);
    MEGA_T(4)
    // ---- stage 2: the middle stage, one token per workgroup
    {
        float *f = (float *)scratch, *red = f + 2176;
        for (int n = wg; n < p.tokens; n += G) {
            pre_mid_body<BF16, 512, true, 16>(p.c1, p.parts, p.tokens, p.bias0, p.descale0, p.gamma1, p.beta1, p.gamma2, p.cosv, p.sinv, p.slotmapping,
                                          p.qscale1, p.qoff1, p.eps, p.q8, p.kv_cache, p.kv_cache_rope, p.per_token ? p.tok0 : nullptr,
                                          p.per_token ? p.tok1 : nullptr, p.cache_mode, p.block_size, p.ctkv_scale, n, f, red);
            __syncthreads();
        }
    }
    MEGA_T(5)
    mega_barrier(p, 2);
    MEGA_T(6)
    // ---- stage 3: per head
    gemm2_bmm_rope_body<BF16, HALF, true>(p.q8, p.tokens, p.wuq, p.q_heads, p.per_token ? nullptr : p.bias1, p.descale1, p.per_token ? p.tok1 : nullptr,
                                          p.wuk_t, p.cosv, p.sinv, p.out0, p.out1, p.q_nope_scale, (int)blockIdx.x, (int)blockIdx.z, lds);
    MEGA_T(7)
}

}  // namespace mi_sgl

using namespace mi_sgl;

extern "C" int mi_mla_pre_gemm_i8_partials(int k) { return k > 0 ? (k + kKC - 1) / kKC : 0; }

extern "C" int mi_mla_pre_gemm_i8(const int8_t *a, int tokens, int k, const int8_t *w, int n, int mode, int32_t *c_i32,
                                  const int32_t *bias, const float *descale, const float *row_scale, void *y, int dtype, void *stream)
{
    if (tokens < 0 || k <= 0 || k % 64 || n <= 0 || (mode != 0 && mode != 1) || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16))
        return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!a || !w || (mode == 0 && !c_i32) || (mode == 1 && (!descale || !y))) return MI_SGL_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int mblocks = (tokens + kBM - 1) / kBM;
    const int chunks = (k + kKC - 1) / kKC;
    if (mode == 0) {
        constexpr int BN = 128;
        static PerDeviceOnce attr_once;                  // 66 KB of dynamic LDS: above the 64 KB default limit
        if (attr_once.need()) {
            (void)hipFuncSetAttribute((const void *)skinny_i8_kernel<0, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BN * kRowStride);
        }
        dim3 grid((n + BN - 1) / BN, chunks, mblocks);
        skinny_i8_kernel<0, BN, true><<<grid, 256, BN * kRowStride, s>>>(a, tokens, k, w, n, c_i32, nullptr, nullptr, nullptr, nullptr);
    } else if (k == kK2) {
        // the op's GEMM2: pick the widest column tile that still gives the chip a full round of workgroups
        static const int force_nt = getenv("MI_MLA_GEMM2_NT") ? atoi(getenv("MI_MLA_GEMM2_NT")) : 0;
        const int nt = force_nt ? force_nt : n >= 256 * 96 ? 6 : n >= 256 * 64 ? 4 : n >= 256 * 32 ? 2 : 1;
        dim3 grid((n + nt * 16 - 1) / (nt * 16), 1, mblocks);
        const size_t lds = (size_t)nt * 16 * kRow2;
#define MI_K1536(NT, B)                                                                                                                  \
        do {                                                                                                                             \
            static PerDeviceOnce attr_once;                                                                                                \
            if (attr_once.need()) {                                                                                                             \
                (void)hipFuncSetAttribute((const void *)skinny_i8_k1536_kernel<NT, B>, hipFuncAttributeMaxDynamicSharedMemorySize, NT * 16 * kRow2); \
            }                                                                                                                            \
            skinny_i8_k1536_kernel<NT, B><<<grid, 256, lds, s>>>(a, tokens, w, n, bias, descale, row_scale, (uint16_t *)y);                          \
        } while (0)
        const bool bf = dtype == MI_DTYPE_BF16;
        if (nt == 6) { if (bf) MI_K1536(6, true); else MI_K1536(6, false); }
        else if (nt == 3) { if (bf) MI_K1536(3, true); else MI_K1536(3, false); }
        else if (nt == 4) { if (bf) MI_K1536(4, true); else MI_K1536(4, false); }
        else if (nt == 2) { if (bf) MI_K1536(2, true); else MI_K1536(2, false); }
        else { if (bf) MI_K1536(1, true); else MI_K1536(1, false); }
#undef MI_K1536
    } else {
        constexpr int BN = 64;
        dim3 grid((n + BN - 1) / BN, 1, mblocks);
        if (dtype == MI_DTYPE_BF16)
            skinny_i8_kernel<1, BN, true><<<grid, 256, BN * kRowStride, s>>>(a, tokens, k, w, n, nullptr, bias, descale, row_scale, (uint16_t *)y);
        else
            skinny_i8_kernel<1, BN, false><<<grid, 256, BN * kRowStride, s>>>(a, tokens, k, w, n, nullptr, bias, descale, row_scale, (uint16_t *)y);
    }
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

#ifdef F_SPAN
extern "C" int mi_dbg_read_f_span(unsigned long long *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_sgl::g_f_span), sizeof(unsigned long long) * 1024 * 2); }
#endif
#ifdef F_TIMING
extern "C" int mi_dbg_read_f(unsigned long long *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_sgl::g_f_dbg), sizeof(unsigned long long) * 4 * 64); }
#endif
extern "C" int mi_mla_pre_bmm_rope(const void *y, int tokens, int q_heads, const void *wuk_t, const void *cos, const void *sin, int dtype,
                                   void *q_out0, void *q_out1, const void *q_nope_scale, void *stream)
{
    if (tokens < 0 || q_heads <= 0 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16)) return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!y || !wuk_t || !cos || !sin || !q_out0 || !q_out1) return MI_SGL_EINVAL;
    dim3 grid(q_heads, 4, (tokens + kBM - 1) / kBM);
    if (dtype == MI_DTYPE_BF16)
        bmm_rope_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>((const uint16_t *)y, tokens, q_heads, (const uint16_t *)wuk_t,
                                                                    (const uint16_t *)cos, (const uint16_t *)sin, (uint16_t *)q_out0,
                                                                    (uint16_t *)q_out1, (const uint16_t *)q_nope_scale);
    else
        bmm_rope_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>((const uint16_t *)y, tokens, q_heads, (const uint16_t *)wuk_t,
                                                                     (const uint16_t *)cos, (const uint16_t *)sin, (uint16_t *)q_out0,
                                                                     (uint16_t *)q_out1, (const uint16_t *)q_nope_scale);
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

extern "C" int mi_mla_pre_gemm2_bmm_rope(const int8_t *a, int tokens, const int8_t *wuq, int q_heads, const int32_t *bias,
                                         const float *descale, const float *row_scale, const void *wuk_t, const void *cos,
                                         const void *sin, int dtype, void *q_out0, void *q_out1, const void *q_nope_scale, void *stream)
{
    if (tokens < 0 || q_heads <= 0 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16)) return MI_SGL_EINVAL;
    if (tokens == 0) return MI_SGL_OK;
    if (!a || !wuq || !descale || !wuk_t || !cos || !sin || !q_out0 || !q_out1) return MI_SGL_EINVAL;
    // 64-row workgroups while 128-row ones would leave CUs idle (and for batches of <= 64 tokens, whose second half would be padding)
    static const int half_env = getenv("MI_MLA_PRE_HALF") ? atoi(getenv("MI_MLA_PRE_HALF")) : -1;
    const bool half = half_env >= 0 ? half_env != 0 : (tokens <= 64 || (long long)q_heads * ((tokens + kBM - 1) / kBM) < 256);
    dim3 grid(q_heads, 1, half ? (tokens + 63) / 64 : (tokens + kBM - 1) / kBM);
#define MI_FUSED(B)                                                                                                                 \
    do {                                                                                                                            \
        static PerDeviceOnce attr_once;                                                                                               \
        if (attr_once.need()) {                                                                                                            \
            (void)hipFuncSetAttribute((const void *)gemm2_bmm_rope_kernel<B, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kF_Lds); \
            (void)hipFuncSetAttribute((const void *)gemm2_bmm_rope_kernel<B, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kF_Lds); \
        }                                                                                                                           \
        if (half)                                                                                                                   \
            gemm2_bmm_rope_kernel<B, true><<<grid, 512, kF_Lds, (hipStream_t)stream>>>(a, tokens, wuq, q_heads, bias, descale, row_scale, \
                                                                             (const uint16_t *)wuk_t, (const uint16_t *)cos,        \
                                                                             (const uint16_t *)sin, (uint16_t *)q_out0,             \
                                                                             (uint16_t *)q_out1, (const uint16_t *)q_nope_scale);   \
        else                                                                                                                        \
            gemm2_bmm_rope_kernel<B, false><<<grid, 512, kF_Lds, (hipStream_t)stream>>>(a, tokens, wuq, q_heads, bias, descale, row_scale, \
                                                                             (const uint16_t *)wuk_t, (const uint16_t *)cos,        \
                                                                             (const uint16_t *)sin, (uint16_t *)q_out0,             \
                                                                             (uint16_t *)q_out1, (const uint16_t *)q_nope_scale);   \
    } while (0)
    if (dtype == MI_DTYPE_BF16) MI_FUSED(true); else MI_FUSED(false);
#undef MI_FUSED
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

// The whole op in one launch (mla_pre_one_launch_kernel above).  -> MI_SGL_OK, or MI_SGL_ENOTAPPLICABLE when the shape is not served
// (more workgroups than the chip runs at once, or than the barrier words hold): the caller then issues the four launches.
extern "C" size_t mi_mla_preprocess_one_launch_sync_words(void) { return (size_t)kMegaBarriers * (kMegaMaxGrid + 16) + 16; }
extern "C" int mi_mla_preprocess_one_launch(const void *hidden, int tokens, int hidden_size, const void *quant_scale0, const int8_t *quant_offset0,
                                            int8_t *a8, float *tok0, const int8_t *wdqkv, int32_t *c1, const int32_t *bias0, const float *descale0,
                                            const void *gamma1, const void *beta1, const void *gamma2, const void *cos, const void *sin,
                                            const int32_t *slotmapping, const void *quant_scale1, const int8_t *quant_offset1, float eps, int8_t *q8,
                                            void *kv_cache, void *kv_cache_rope, float *tok1, int cache_mode, int block_size, const void *ctkv_scale,
                                            const int8_t *wuq, int q_heads, const int32_t *bias1, const float *descale1, const void *wuk_t,
                                            void *q_out0, void *q_out1, const void *q_nope_scale, int per_token, int dtype, uint32_t *sync_words,
                                            void *stream)
{
    if (tokens <= 0 || q_heads <= 0 || hidden_size <= 0 || hidden_size % 64 || (dtype != MI_DTYPE_BF16 && dtype != MI_DTYPE_F16) || !sync_words)
        return MI_SGL_EINVAL;
    if (!hidden || !a8 || !wdqkv || !c1 || !descale0 || !gamma1 || !beta1 || !gamma2 || !cos || !sin || !slotmapping || !q8 || !kv_cache ||
        !kv_cache_rope || !wuq || !descale1 || !wuk_t || !q_out0 || !q_out1 || (per_token ? (!tok0 || !tok1) : (!quant_scale0 || !quant_offset0 || !quant_scale1 || !quant_offset1)) ||
        cache_mode < 1 || cache_mode > 3 || (cache_mode != 1 && block_size <= 0) || (cache_mode == 2 && !ctkv_scale))
        return MI_SGL_EINVAL;
    static const int half_env = getenv("MI_MLA_PRE_HALF") ? atoi(getenv("MI_MLA_PRE_HALF")) : -1;
    const bool half = half_env >= 0 ? half_env != 0 : (tokens <= 64 || (long long)q_heads * ((tokens + kBM - 1) / kBM) < 256);
    const int gz = half ? (tokens + 63) / 64 : (tokens + kBM - 1) / kBM;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    // every workgroup waits for every other one at the barriers: all of them must be resident at once (one per CU: the ring takes the LDS)
    if ((long long)q_heads * gz > cus || (long long)q_heads * gz > kMegaMaxGrid) return MI_SGL_ENOTAPPLICABLE;
    MegaArgs p{};
    p.x = (const uint16_t *)hidden, p.tokens = tokens, p.hidden = hidden_size, p.qscale0 = (const uint16_t *)quant_scale0, p.qoff0 = quant_offset0;
    p.a8 = a8, p.tok0 = tok0, p.wdqkv = wdqkv, p.c1 = c1, p.parts = (hidden_size + kKC - 1) / kKC;
    p.bias0 = bias0, p.descale0 = descale0, p.gamma1 = (const uint16_t *)gamma1, p.beta1 = (const uint16_t *)beta1, p.gamma2 = (const uint16_t *)gamma2;
    p.cosv = (const uint16_t *)cos, p.sinv = (const uint16_t *)sin, p.slotmapping = slotmapping, p.qscale1 = (const uint16_t *)quant_scale1;
    p.qoff1 = quant_offset1, p.eps = eps, p.q8 = q8, p.kv_cache = (uint16_t *)kv_cache, p.kv_cache_rope = (uint16_t *)kv_cache_rope, p.tok1 = tok1;
    p.cache_mode = cache_mode, p.block_size = block_size, p.ctkv_scale = (const uint16_t *)ctkv_scale;
    p.wuq = wuq, p.q_heads = q_heads, p.bias1 = bias1, p.descale1 = descale1, p.wuk_t = (const uint16_t *)wuk_t, p.out0 = (uint16_t *)q_out0;
    p.out1 = (uint16_t *)q_out1, p.q_nope_scale = (const uint16_t *)q_nope_scale, p.flags = sync_words, p.epoch = 0, p.per_token = per_token ? 1 : 0;
    dim3 grid(q_heads, 1, gz);
#define MI_MEGA(B, H)                                                                                                              \
    do {                                                                                                                            \
        static PerDeviceOnce attr_once;                                                                                             \
        if (attr_once.need())                                                                                                       \
            (void)hipFuncSetAttribute((const void *)mla_pre_one_launch_kernel<B, H>, hipFuncAttributeMaxDynamicSharedMemorySize, kF_Lds); \
        mla_pre_one_launch_kernel<B, H><<<grid, 512, kF_Lds, (hipStream_t)stream>>>(p);                                             \
    } while (0)
    if (dtype == MI_DTYPE_BF16) { if (half) MI_MEGA(true, true); else MI_MEGA(true, false); }
    else { if (half) MI_MEGA(false, true); else MI_MEGA(false, false); }
#undef MI_MEGA
    return hipGetLastError() == hipSuccess ? MI_SGL_OK : MI_SGL_ELAUNCH;
}

#ifdef MEGA_TIMING
extern "C" int mi_dbg_read_mega(unsigned long long *dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mi_sgl::g_mega_t), sizeof(unsigned long long) * 1024 * 8); }
#endif
