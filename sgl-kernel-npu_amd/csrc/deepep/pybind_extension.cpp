// pybind11 module `deep_ep_cpp` for MI355X.  It exposes the classes / method names / positional argument orders that
// python/deep_ep (and SGLang behind it) call on the reference module (csrc/deepep/pybind_extension.cpp:17-55), plus the
// hipIpc window bootstrap and the pack / unpack entry points of the RCCL `alltoall` strategies.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <torch/extension.h>

#include "deep_ep.hpp"

#ifndef TORCH_EXTENSION_NAME
#define TORCH_EXTENSION_NAME deep_ep_cpp
#endif

namespace py = pybind11;
using deep_ep::Buffer;

// one member function under its own name
#define MI_METHOD(cls, name) cls.def(#name, &Buffer::name)

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X (gfx950) DeepEP host runtime";
    py::register_exception<deep_ep::EPException>(m, "EPException", PyExc_RuntimeError);
    m.def("get_low_latency_rdma_size_hint", &deep_ep::get_low_latency_rdma_size_hint);

    // ---- API-compat value types (reference config.hpp:10-35, event.hpp:6-15)
    py::class_<deep_ep::Config> config(m, "Config");
    config.def(py::init<int, int, int, int, int>(), py::arg("num_sms") = 20, py::arg("num_max_nvl_chunked_send_tokens") = 6,
               py::arg("num_max_nvl_chunked_recv_tokens") = 256, py::arg("num_max_rdma_chunked_send_tokens") = 6,
               py::arg("num_max_rdma_chunked_recv_tokens") = 256);
    config.def_readonly("num_sms", &deep_ep::Config::num_sms);
    config.def("get_nvl_buffer_size_hint", &deep_ep::Config::get_nvl_buffer_size_hint);
    config.def("get_rdma_buffer_size_hint", &deep_ep::Config::get_rdma_buffer_size_hint);

    py::class_<deep_ep::EventHandle> event(m, "EventHandle");
    event.def(py::init<>());
    event.def("current_stream_wait", &deep_ep::EventHandle::current_stream_wait);

    py::class_<Buffer> buf(m, "Buffer");
    buf.def(py::init<int, int, int64_t, int64_t, bool, std::string>());

    // ---- MI355X only: symmetric-window bootstrap over hipIpc (replaces the HCCL window lookup by communicator name)
    MI_METHOD(buf, get_local_device_id);
    MI_METHOD(buf, get_local_device_bus_id);
    MI_METHOD(buf, set_ranks_share_device);
    MI_METHOD(buf, get_ranks_share_device);
    MI_METHOD(buf, get_local_window_ptrs);
    MI_METHOD(buf, get_window_bytes);
    buf.def("get_local_ipc_handle", [](const Buffer &b) { return py::bytes(b.get_local_ipc_handle()); });
    buf.def("sync", &Buffer::sync, py::arg("handles"), py::arg("local_ptrs"));

    // ---- queries
    MI_METHOD(buf, is_available);
    MI_METHOD(buf, is_window_fine_grained);
    MI_METHOD(buf, self_test);
    MI_METHOD(buf, set_fused_requant);
    MI_METHOD(buf, get_fused_requant);
    MI_METHOD(buf, get_gemm_xcds);
    MI_METHOD(buf, self_test_in_launch);
    MI_METHOD(buf, set_two_launch_forms);
    MI_METHOD(buf, get_two_launch_forms);
    MI_METHOD(buf, get_low_latency_launch_forms);
    MI_METHOD(buf, get_low_latency_default_forms);
    MI_METHOD(buf, set_dispatch_transport);
    MI_METHOD(buf, get_dispatch_transport);
    MI_METHOD(buf, set_local_row_paths);
    MI_METHOD(buf, get_local_row_paths);
    MI_METHOD(buf, set_fused_rows_in_place);
    MI_METHOD(buf, get_fused_rows_in_place);
    MI_METHOD(buf, get_num_rdma_ranks);
    MI_METHOD(buf, get_rdma_rank);
    MI_METHOD(buf, get_notify_send_data);

    // ---- normal (high-throughput) mode
    MI_METHOD(buf, get_dispatch_layout);
    MI_METHOD(buf, intranode_dispatch);
    MI_METHOD(buf, intranode_combine);
    MI_METHOD(buf, notify_verify);
    for (const char *name : {"internode_dispatch", "internode_combine"})      // one xGMI node is a single rdma rank
        buf.def(name, [](Buffer &b, py::args, py::kwargs) { b.internode_unsupported(); });

    // ---- low-latency mode and the fused MoE paths
    MI_METHOD(buf, clean_low_latency_buffer);
    MI_METHOD(buf, low_latency_dispatch);
    MI_METHOD(buf, low_latency_combine);
    MI_METHOD(buf, dispatch_ffn_combine);
    MI_METHOD(buf, clear_weight_cache);
    buf.def("fused_deep_moe", &Buffer::fused_deep_moe, py::arg("x"), py::arg("expert_ids"), py::arg("gmm1_permuted_weight"),
            py::arg("gmm1_permuted_weight_scale"), py::arg("gmm2_weight"), py::arg("gmm2_weight_scale"),
            py::arg("expert_scales_optional"), py::arg("num_max_dispatch_tokens_per_rank"), py::arg("num_experts"),
            py::arg("quant_mode"), py::arg("profile_enable") = false);

    // ---- per-kernel profiler (HIP events on the caller's stream)
    buf.def("begin_profile", &Buffer::begin_profile, py::arg("num_profile_skip_launches"), py::arg("num_profile_active_launches"),
            py::arg("profile_trace_dir") = "");
    MI_METHOD(buf, end_profile);
    MI_METHOD(buf, get_profile_summary);
    MI_METHOD(buf, get_gemm_clock);

    // ---- MI355X only: pack / unpack kernels behind the RCCL `alltoall` strategies
    MI_METHOD(buf, a2a_dispatch_stage);
    MI_METHOD(buf, a2a_dispatch_tables);
    MI_METHOD(buf, a2a_dispatch_unpack);
    MI_METHOD(buf, a2a_combine_pack);
    MI_METHOD(buf, a2a_combine_prepare);
    MI_METHOD(buf, a2a_combine_reduce);
}
