// hipFuncSetAttribute (the dynamic-LDS limit of a kernel) is per DEVICE: a process-wide `static bool` raises it on the first device only,
// and a single-process multi-GPU caller's launches on the others fail with hipErrorInvalidValue.  One of these per call site:
//     static PerDeviceOnce once;  if (once.need()) { hipFuncSetAttribute(...); }
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    bool need()
    {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return true;
        const uint64_t bit = 1ull << (d & 63);
        return !(done.fetch_or(bit, std::memory_order_relaxed) & bit);
    }
};
