// Scratch microbenchmark: what does a dependent kernel launch cost on this GPU, and what does one agent-scope hop between two XCDs cost?
//   (a) N empty kernels back to back on one stream (1 workgroup; 256 workgroups x 512 threads with 150 KB of LDS each)
//   (b) the same with hipGraph replay
//   (c) ping-pong of a flag between workgroup 0 and workgroup 1 (round-robin placement puts them on different XCDs): relaxed stores / loads
//       at agent scope, and the same with a release fence before every store and an acquire fence after every successful poll
//   (d) the same ping-pong while 236 other workgroups poll a third word in the same 256-byte line (what a grid barrier looks like)
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor && ./launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ __launch_bounds__(512) void big_empty_kernel(int *p)
{
    extern __shared__ uint8_t lds[];
    if (p && threadIdx.x == 9999) *p = lds[0];
}

template <bool FENCES>
__global__ void pingpong(uint32_t *flags, int rounds, int crowd, uint64_t *ticks)
{
    uint32_t *a = flags, *b = flags + 1, *c = flags + 2, *stop = flags + 3;
    if (threadIdx.x != 0) return;
    if (blockIdx.x == 0) {
        const uint64_t t0 = wall_clock64();
        for (int i = 1; i <= rounds; ++i) {
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(a, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) {}
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        ticks[0] = wall_clock64() - t0;
        __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (blockIdx.x == 1) {
        for (int i = 1; i <= rounds; ++i) {
            while (__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)i) {}
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (FENCES) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(b, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if ((int)blockIdx.x < 2 + crowd) {                 // the crowd: polls a neighbouring word until the ping-pong is over
        while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            (void)__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_sleep(1);
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)big_empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int N = 2000;
    for (int shape = 0; shape < 2; ++shape) {
        auto launch = [&]() {
            if (shape == 0) empty_kernel<<<1, 64, 0, s>>>(nullptr);
            else big_empty_kernel<<<256, 512, 150 * 1024, s>>>(nullptr);
        };
        for (int i = 0; i < 200; ++i) launch();
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < N; ++i) launch();
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per launch back to back (stream)\n", shape ? "256 x 512 threads, 150 KB LDS" : "1 x 64 threads", ms * 1e3 / N);
        hipGraph_t g;
        hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 100; ++i) launch();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.2f us per launch in a replayed graph of 100\n", shape ? "256 x 512 threads, 150 KB LDS" : "1 x 64 threads", ms * 1e3 / 2000);
    }
    uint32_t *flags;
    uint64_t *ticks, h;
    hipMalloc(&flags, 256);
    hipMalloc(&ticks, 8);
    const int rounds = 2000;
    for (int fences = 0; fences < 2; ++fences)
        for (int crowd : {0, 236}) {
            hipMemsetAsync(flags, 0, 256, s);
            if (fences) pingpong<true><<<2 + crowd, 64, 0, s>>>(flags, rounds, crowd, ticks);
            else pingpong<false><<<2 + crowd, 64, 0, s>>>(flags, rounds, crowd, ticks);
            hipStreamSynchronize(s);
            hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
            printf("flag hop between two workgroups (%s, %d workgroups polling next to it): %.2f us\n", fences ? "release + acquire fences" : "relaxed",
                   crowd, (double)h / 100.0 / (2.0 * rounds));
        }
    (void)now_us;
    return 0;
}
