// Scratch microbenchmark: what bounds the QK^T phase of the eight-wave MLA kernel -- LDS operand bandwidth, LDS latency or the matrix pipe?
// 512-thread workgroups (two waves per SIMD and one workgroup per CU, like the kernel),
// each wave runs `iters` phases of N MFMAs with one 1-KiB ds_read_b128 operand fragment per MFMA from a 32-key x 1056-B-stride tile,
// operand ring `AHEAD` deep.  Modes: both / reads only / MFMAs only, 16x16x32 (36 per phase, every wave reads the whole tile) and
// 32x32x16 (18 per phase: the k-split form, half the operand bytes per FLOP), 8 or 4 active waves.
//   hipcc --offload-arch=gfx950 -O3 lds_mfma_rate.hip -o lds_mfma_rate && ./lds_mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStride = 1056;

// MODE bit 0: reads, bit 1: MFMAs.  BIG: 32x32x16 (18 steps) instead of 16x16x32 (36 steps).
template <int MODE, bool BIG, int AHEAD>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void phase_kernel(int iters, int active_waves, float *sink, uint64_t *cycles)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960; i += 512) ((uint32_t *)lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    if (wave >= active_waves) return;
    const int h16 = lane & 15, g = lane >> 4, c32 = lane & 31, kg = lane >> 5;
    const uint8_t *abase = BIG ? lds + c32 * kStride + ((kg ^ ((c32 >> 3) & 1)) * 16) : lds + h16 * kStride + g * 16;      // (both conflict-free)
    constexpr int N = BIG ? 18 : 36;
    s16x8 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = s16x8{(short)(lane + i), 1, 2, 3, 4, 5, 6, 7};
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    f32x16 sb = {0};
    s16x8 keep = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        s16x8 af[AHEAD + 1];
        auto lda = [&](int step) -> s16x8 {
            if constexpr (BIG) return *(const s16x8 *)(abase + (step & 1) * 32 + (step >> 1) * 64);      // 32 keys x 16 dims per step
            else return *(const s16x8 *)(abase + (step & 1) * 16 * kStride + (step >> 1) * 64);
        };
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE & 1)
#pragma unroll
            for (int p = 0; p < AHEAD; ++p) af[p] = lda(p);
#pragma unroll
        for (int step = 0; step < N; ++step) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 1)
                if (step + AHEAD < N) af[(step + AHEAD) % (AHEAD + 1)] = lda(step + AHEAD);
            __builtin_amdgcn_sched_barrier(0);
            s16x8 a;
            if constexpr (MODE & 1) a = af[step % (AHEAD + 1)];
            else a = q[(step + 3) & 7];
            if constexpr (MODE & 2) {
                if constexpr (BIG) sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, q[step & 7]), sb, 0, 0, 0);
                else if (step & 1) s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, q[(step >> 1) & 7]), s1, 0, 0, 0);
                else s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, q[(step >> 1) & 7]), s0, 0, 0, 0);
            } else {
                keep ^= a;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float r = s0[0] + s1[1] + sb[3] + (float)keep[0];
    if (r == 1234.5f) sink[threadIdx.x] = r;
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, bool BIG, int AHEAD>
void run(const char *name, int wgs, int active, float *sink, uint64_t *cyc)
{
    const int iters = 2000;
    hipFuncSetAttribute((const void *)phase_kernel<MODE, BIG, AHEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
    for (int rep = 0; rep < 2; ++rep) phase_kernel<MODE, BIG, AHEAD><<<wgs, 512, 163840>>>(iters, active, sink, cyc);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(wgs * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int w = 0; w < wgs; ++w)
        for (int i = 0; i < active; ++i) sum += (double)h[w * 8 + i];
    const double per_phase = sum / (wgs * active) / iters;
    const int n = BIG ? 18 : 36;
    const double bytes = (MODE & 1) ? (double)active * n * 1024 : 0;
    printf("%-44s wgs %3d waves %d ahead %d: %7.0f cycles / phase  (%5.1f per step)  LDS %6.1f B/clk/CU\n", name, wgs, active, AHEAD, per_phase, per_phase / n,
           bytes / per_phase);
}

int main()
{
    float *sink;
    uint64_t *cyc;
    hipMalloc(&sink, 4096);
    hipMalloc(&cyc, 256 * 8 * 8);
    for (int wgs : {1, 256}) {
        run<3, false, 2>("16x16x32 reads + mfma", wgs, 8, sink, cyc);
        run<3, false, 6>("16x16x32 reads + mfma", wgs, 8, sink, cyc);
        run<1, false, 6>("16x16x32 reads only", wgs, 8, sink, cyc);
        run<2, false, 6>("16x16x32 mfma only", wgs, 8, sink, cyc);
        run<3, false, 6>("16x16x32 reads + mfma", wgs, 4, sink, cyc);
        run<1, false, 6>("16x16x32 reads only", wgs, 4, sink, cyc);
        run<2, false, 6>("16x16x32 mfma only", wgs, 4, sink, cyc);
        run<3, true, 4>("32x32x16 reads + mfma (k-split)", wgs, 8, sink, cyc);
        run<1, true, 4>("32x32x16 reads only", wgs, 8, sink, cyc);
        run<2, true, 4>("32x32x16 mfma only", wgs, 8, sink, cyc);
        run<3, true, 4>("32x32x16 reads + mfma (k-split)", wgs, 4, sink, cyc);
    }
    return 0;
}
