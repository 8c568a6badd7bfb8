// Scratch probe: how fast can a CU stream HBM-resident bytes?  Modes: 0 = vector loads into registers, 1 = LDS-DMA, 2 = both at once.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_probe tools/probes/stream_probe.hip && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(uint32_t dst, const void *vaddr)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(vaddr) : "memory");
}

// each workgroup streams `bytes_per_wg` contiguous bytes; waves x 64 lanes x 16 B per request, `depth` requests in flight per wave
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const uint8_t *src, size_t bytes_per_wg, uint32_t *sink)
{
    extern __shared__ uint8_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint8_t *base = src + (size_t)blockIdx.x * bytes_per_wg;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)lds);
    u32x4 acc = {0, 0, 0, 0};
    const size_t step = (size_t)nw * 1024;                 // bytes per "row" of requests across the waves
    const size_t iters = bytes_per_wg / step;
    for (size_t i = 0; i < iters; i += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint8_t *p = base + (i + d) * step + wave * 1024 + lane * 16;
            if (MODE == 0 || (MODE == 2 && (d & 1))) v[d] = *(const u32x4 *)p;
            else dma16(lds_base + (uint32_t)((wave * DEPTH + d) * 1024), p);
        }
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (MODE == 0 || (MODE == 2 && (d & 1))) acc ^= v[d];
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

// write side: each workgroup stores `bytes_per_wg` contiguous bytes, 16 B per lane; NT = nontemporal stores
template <bool NT>
__global__ __launch_bounds__(512) void store_kernel(uint8_t *dst, size_t bytes_per_wg)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint8_t *base = dst + (size_t)blockIdx.x * bytes_per_wg;
    const size_t step = (size_t)nw * 1024, iters = bytes_per_wg / step;
    const u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
    for (size_t i = 0; i < iters; ++i) {
        u32x4 *p = (u32x4 *)(base + i * step + wave * 1024 + lane * 16);
        if (NT) __builtin_nontemporal_store(v, p);
        else *p = v;
    }
}

template <bool NT>
static void run_store(uint8_t *dst, size_t total, int wgs, int threads)
{
    const size_t per = (total / wgs) / (threads / 64 * 1024) * (threads / 64 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        store_kernel<NT><<<wgs, threads>>>(dst, per);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("store nt %d wgs %4d threads %4d (%6.1f MB): %7.1f us  %6.2f TB/s  %6.1f GB/s per WG\n", (int)NT, wgs, threads, per * wgs / 1e6,
           best * 1e3, per * wgs / (best * 1e-3) / 1e12, per / (best * 1e-3) / 1e9);
}

// the access pattern of combine_reduce: a wave takes one 1 KB segment of 8 pseudo-random 14 KB rows (a token's K expert rows)
__global__ __launch_bounds__(256) void gather_kernel(const uint8_t *src, size_t nrows, size_t tokens, uint32_t *sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t t = wid / 14, seg = wid % 14;
    if (t >= tokens) return;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const size_t row = ((t * 8 + k) * 2654435761ull + 12345) % nrows;
        v[k] = __builtin_nontemporal_load((const u32x4 *)(src + row * 14336 + seg * 1024 + lane * 16));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k];
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

template <int MODE, int DEPTH>
static void run(const uint8_t *src, size_t total, int wgs, int threads, uint32_t *sink)
{
    const size_t per = total / wgs / (threads / 64 * 1024 * DEPTH) * (threads / 64 * 1024 * DEPTH);
    hipFuncSetAttribute((const void *)stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = (size_t)threads / 64 * DEPTH * 1024;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        stream_kernel<MODE, DEPTH><<<wgs, threads, lds>>>(src, per, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("mode %d depth %2d wgs %4d threads %4d: %7.1f us  %6.2f TB/s  %6.1f GB/s per WG\n", MODE, DEPTH, wgs, threads, best * 1e3,
           per * wgs / (best * 1e-3) / 1e12, per / (best * 1e-3) / 1e9);
}

int main()
{
    const size_t total = 1ull << 30;                       // 1 GiB, far beyond L2 + MALL
    uint8_t *src;
    uint32_t *sink;
    hipMalloc(&src, total), hipMalloc(&sink, 64);
    hipMemset(src, 1, total);
    for (int wgs : {128, 256, 512}) {
        run<0, 4>(src, total, wgs, 512, sink);
        run<0, 8>(src, total, wgs, 512, sink);
        run<1, 4>(src, total, wgs, 512, sink);
        run<1, 8>(src, total, wgs, 512, sink);
        run<1, 16>(src, total, wgs, 512, sink);
        run<2, 8>(src, total, wgs, 512, sink);
        run<2, 16>(src, total, wgs, 512, sink);
    }
    // per-CU store rate: few workgroups (one per CU at most), 4 MB each, then the whole chip
    for (int wgs : {8, 32, 128, 256, 1024}) {
        const size_t bytes = (size_t)wgs * (4u << 20) < total ? (size_t)wgs * (4u << 20) : total;      // never past the allocation
        run_store<false>(src, bytes, wgs, 512);
        run_store<true>(src, bytes, wgs, 512);
    }
    run_store<false>(src, total, 2048, 256);
    run_store<true>(src, total, 2048, 256);
    run<0, 8>(src, total, 2048, 256, sink);
    run<1, 8>(src, total, 2048, 256, sink);
    {
        const size_t nrows = total / 14336, tokens = 4096 * 2;      // 8192 tokens x 8 rows x 14 KB = 0.94 GB of reads
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0);
            gather_kernel<<<(unsigned)((tokens * 14 + 3) / 4), 256>>>(src, nrows, tokens, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("gather of 8 random 14 KB rows per token, 1 KB per wave and row: %7.1f us  %6.2f TB/s\n", best * 1e3,
               tokens * 8 * 14336.0 / (best * 1e-3) / 1e12);
    }
    return 0;
}
