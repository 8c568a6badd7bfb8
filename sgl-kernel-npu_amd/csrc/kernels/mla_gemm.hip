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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mi_sgl_kernels.h"

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

// MODE 0: atomic int32 accumulate (split-K, blockIdx.y = chunk);  MODE 1: whole K, dequant epilogue into the I/O dtype
template <int MODE, int BN, bool BF16>
__global__ __launch_bounds__(256) void skinny_i8_kernel(const int8_t *__restrict__ A, int M, int K, const int8_t *__restrict__ W, int N,
                                                       int32_t *__restrict__ C, const int32_t *__restrict__ bias,
                                                       const float *__restrict__ descale, const float *__restrict__ row_scale,
                                                       uint16_t *__restrict__ Y)
{
    constexpr int NT = BN / 16;                       // MFMA column tiles, all of them handled by every wave
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // [BN][kRowStride]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.z * kBM + wave * 32;
    const int chunks = (K + kKC - 1) / kKC;
    const int c_begin = MODE == 0 ? blockIdx.y : 0, c_end = MODE == 0 ? blockIdx.y + 1 : chunks;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);

    i32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = i32x4{0, 0, 0, 0};

    const int8_t *arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) arow[mt] = A + (size_t)min(m0 + mt * 16 + c16, M - 1) * K + g * 16;      // rows past M: any valid row

    for (int c = c_begin; c < c_end; ++c) {
        const int k0 = c * kKC;
        const int klen = min(kKC, K - k0);            // multiple of 64
        // weights: wave w moves rows w*BN/4 ..; lane l < klen/16 carries 16 B of the row's chunk
#pragma unroll 4
        for (int r = 0; r < BN / 4; ++r) {
            const int row = wave * (BN / 4) + r;
            const int8_t *src = W + (size_t)min(n0 + row, N - 1) * K + k0 + lane * 16;
            if (lane * 16 < klen) dma16(lds_base + (uint32_t)(row * kRowStride), src);
        }
        // activations of this wave's 32 token rows, straight into MFMA operand layout (L2-resident)
        // Every column tile of a K-chunk reads the SAME activation lines; started in the same order by every workgroup they all
        // queue on one L2 channel at a time.  Workgroup x starts its sweep x k-steps into the chunk (and wraps).
        const int nks = klen / 64;
        const int rot = blockIdx.x % nks;
        i32x4 af[2][kKC / 64];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
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
                acc[0][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[0][ks], bf, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[1][ks], bf, acc[1][nt], 0, 0, 0);
            }
        }
        __syncthreads();                              // everybody is done with the stage before the next chunk lands
    }

    // lane holds C[row = m0 + mt*16 + 4g + r][col = n0 + nt*16 + c16]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
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
                    C[((size_t)blockIdx.y * M + row) * N + col] = acc[mt][nt][r];
                } else {
                    float y = (float)(acc[mt][nt][r] + (bias ? bias[col] : 0)) * descale[col];
                    if (row_scale) y = y * row_scale[row];      // per_token_quant_symm: the token's own scale (hpp:2273-2279)
                    Y[(size_t)row * N + col] = sth16<BF16>(y);
                }
            }
        }
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
        static bool attr_set = false;                  // 66 KB of dynamic LDS: above the 64 KB default limit
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)skinny_i8_kernel<0, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, BN * kRowStride);
            attr_set = true;
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
            static bool attr_set = false;                                                                                                \
            if (!attr_set) {                                                                                                             \
                (void)hipFuncSetAttribute((const void *)skinny_i8_k1536_kernel<NT, B>, hipFuncAttributeMaxDynamicSharedMemorySize, NT * 16 * kRow2); \
                attr_set = true;                                                                                                         \
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
