// API-compatibility shells (reference csrc/deepep/config.hpp:10-35, config.cpp:4-24, event.hpp:6-15).
#pragma once
#include <cstdlib>
#include <string>

#include <hip/hip_runtime_api.h>

namespace deep_ep {

// DeepEP tuning knobs; on MI355X only num_sms % 2 == 0 is checked (as in the reference, deep_ep.cpp:210).
struct Config {
    int num_sms;
    int num_max_nvl_chunked_send_tokens;
    int num_max_nvl_chunked_recv_tokens;
    int num_max_rdma_chunked_send_tokens;
    int num_max_rdma_chunked_recv_tokens;

    Config(int sms, int a, int b, int c, int d)
        : num_sms(sms),
          num_max_nvl_chunked_send_tokens(a),
          num_max_nvl_chunked_recv_tokens(b),
          num_max_rdma_chunked_send_tokens(c),
          num_max_rdma_chunked_recv_tokens(d)
    {}
    // size hints return their first argument, like the reference
    size_t get_nvl_buffer_size_hint(size_t hidden_bytes, int) const { return hidden_bytes; }
    size_t get_rdma_buffer_size_hint(int64_t hidden_bytes, int) const { return (size_t)hidden_bytes; }
};

inline size_t get_low_latency_rdma_size_hint(int num_max_dispatch_tokens_per_rank, int, int, int)
{
    return (size_t)num_max_dispatch_tokens_per_rank;
}

inline int get_value_from_env(const std::string &name, int default_value)
{
    const char *v = std::getenv(name.c_str());
    if (!v || !*v) return default_value;
    char *end = nullptr;
    long x = std::strtol(v, &end, 10);
    return (end && *end == '\0') ? (int)x : default_value;
}

inline long long get_ll_from_env(const char *name, long long default_value)
{
    const char *v = std::getenv(name);
    if (!v || !*v) return default_value;
    char *end = nullptr;
    long long x = std::strtoll(v, &end, 10);
    return (end && *end == '\0') ? x : default_value;
}

// The reference's EventHandle is an empty shell whose wait is a no-op because every op runs on the caller's
// stream.  Ours does the same work on the caller's stream, so stream order already covers it; the handle still
// records a real HIP event so `current_stream_wait()` is correct if the caller switched streams.
struct EventHandle {
    hipEvent_t ev = nullptr;
    EventHandle();
    EventHandle(const EventHandle &) = default;
    void current_stream_wait() const;
};

}  // namespace deep_ep
