// Scratch microbenchmark: how many bytes per clock can ONE CU pull from L2 / HBM?  512-thread workgroups, each streams its own region
// with global_load_dwordx4 (K loads in flight per wave) or with global_load_lds_dwordx4; grid = number of CUs to load.
//   hipcc --offload-arch=gfx950 -O3 cu_fetch_rate.hip -o cu_fetch_rate && ./cu_fetch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int K>
__global__ __launch_bounds__(512) void stream_regs(const uint4 *src, size_t region_u4, int iters, uint4 *sink, uint64_t *cycles)
{
    const uint4 *base = src + (size_t)blockIdx.x * region_u4;
    const int tid = threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    size_t pos = tid;
    for (int it = 0; it < iters; ++it) {
        uint4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v[k] = base[pos];
            pos += blockDim.x;
            if (pos >= region_u4) pos -= region_u4;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc.x ^= v[k].x, acc.y ^= v[k].y, acc.z ^= v[k].z, acc.w ^= v[k].w;
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (acc.x == 0x12345678u) sink[tid] = acc;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int K>
__global__ __launch_bounds__(512) void stream_dma(const uint4 *src, size_t region_u4, int iters, uint4 *sink, uint64_t *cycles)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint4 *base = src + (size_t)blockIdx.x * region_u4;
    const int tid = threadIdx.x, wave = tid >> 6;
    const uint32_t lbase = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds + wave * (K * 1024));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    size_t pos = tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint4 *a = base + pos;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lbase + k * 1024), "v"(a) : "memory", "m0");
            pos += blockDim.x;
            if (pos >= region_u4) pos -= region_u4;
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (lds[tid] == 0x77 && iters < 0) sink[tid] = uint4{1, 2, 3, 4};
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    const size_t region = 4u << 20;                 // 4 MiB per workgroup
    const int max_wgs = 256;
    uint4 *src, *sink;
    uint64_t *cyc;
    hipMalloc(&src, region * max_wgs);
    hipMalloc(&sink, 512 * 16);
    hipMalloc(&cyc, max_wgs * 8);
    hipMemset(src, 1, region * max_wgs);
    std::vector<uint64_t> h(max_wgs);
    auto report = [&](const char *name, int wgs, size_t bytes_per_wg, float ms) {
        hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < wgs; ++i) s += (double)h[i];
        s /= wgs;
        printf("%-28s wgs %3d  %.1f B/clk/CU  (%.0f cycles)  chip %.2f TB/s\n", name, wgs, bytes_per_wg / s, s, wgs * (double)bytes_per_wg / (ms * 1e-3) / 1e12);
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int region_kb : {256, 4096}) {               // 256 KiB per WG: L2-resident after the first pass; 4 MiB: streams from HBM
        const size_t ru4 = (size_t)region_kb * 1024 / 16;
        printf("region per workgroup %d KiB\n", region_kb);
        for (int threads : {64, 128, 256}) {
            const int wgs = 8;
#define RUNT(NAME, KERNEL, K, LDSB)                                                                \
    {                                                                                              \
        const int iters = (int)((4u << 20) / (64 * 16 * K));                                       \
        hipLaunchKernelGGL(KERNEL, dim3(wgs), dim3(threads), LDSB, 0, src, ru4, iters, sink, cyc); \
        hipEventRecord(e0);                                                                        \
        hipLaunchKernelGGL(KERNEL, dim3(wgs), dim3(threads), LDSB, 0, src, ru4, iters, sink, cyc); \
        hipEventRecord(e1);                                                                        \
        hipEventSynchronize(e1);                                                                   \
        float ms;                                                                                  \
        hipEventElapsedTime(&ms, e0, e1);                                                          \
        printf("threads %d: ", threads);                                                           \
        report(NAME, wgs, (size_t)iters * threads * 16 * K, ms);                                   \
    }
            RUNT("regs, 16 in flight / wave", stream_regs<16>, 16, 0)
            RUNT("lds-dma, 16 in flight / wave", stream_dma<16>, 16, 8 * 16 * 1024)
        }
        for (int wgs : {1, 8, 64, 256}) {
            const int iters8 = (int)((32u << 20) / (512 * 16 * 8));        // 32 MiB per WG
#define RUN(NAME, KERNEL, K, LDSB)                                                                 \
    {                                                                                              \
        const int iters = (int)((32u << 20) / (512 * 16 * K));                                     \
        hipLaunchKernelGGL(KERNEL, dim3(wgs), dim3(512), LDSB, 0, src, ru4, iters, sink, cyc);     \
        hipEventRecord(e0);                                                                        \
        hipLaunchKernelGGL(KERNEL, dim3(wgs), dim3(512), LDSB, 0, src, ru4, iters, sink, cyc);     \
        hipEventRecord(e1);                                                                        \
        hipEventSynchronize(e1);                                                                   \
        float ms;                                                                                  \
        hipEventElapsedTime(&ms, e0, e1);                                                          \
        report(NAME, wgs, (size_t)iters * 512 * 16 * K, ms);                                       \
    }
            (void)iters8;
            RUN("regs, 4 in flight / wave", stream_regs<4>, 4, 0)
            RUN("regs, 8 in flight / wave", stream_regs<8>, 8, 0)
            RUN("regs, 16 in flight / wave", stream_regs<16>, 16, 0)
            RUN("lds-dma, 8 in flight / wave", stream_dma<8>, 8, 8 * 8 * 1024)
            RUN("lds-dma, 16 in flight / wave", stream_dma<16>, 16, 8 * 16 * 1024)
        }
    }
    return 0;
}
