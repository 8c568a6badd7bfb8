// pybind11 module `deep_ep_cpp` for MI355X.  Same classes / methods / argument orders as the reference module
// (csrc/deepep/pybind_extension.cpp:17-55) plus the window bootstrap and the alltoall-strategy kernel entry points.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <torch/extension.h>

#include "deep_ep.hpp"

#ifndef TORCH_EXTENSION_NAME
#define TORCH_EXTENSION_NAME deep_ep_cpp
#endif

namespace py = pybind11;

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.doc() = "MI355X (gfx950) DeepEP host runtime";
    py::register_exception<deep_ep::EPException>(m, "EPException", PyExc_RuntimeError);

    py::class_<deep_ep::Config>(m, "Config")
        .def(py::init<int, int, int, int, int>(), py::arg("num_sms") = 20, py::arg("num_max_nvl_chunked_send_tokens") = 6,
             py::arg("num_max_nvl_chunked_recv_tokens") = 256, py::arg("num_max_rdma_chunked_send_tokens") = 6,
             py::arg("num_max_rdma_chunked_recv_tokens") = 256)
        .def_readonly("num_sms", &deep_ep::Config::num_sms)
        .def("get_nvl_buffer_size_hint", &deep_ep::Config::get_nvl_buffer_size_hint)
        .def("get_rdma_buffer_size_hint", &deep_ep::Config::get_rdma_buffer_size_hint);
    m.def("get_low_latency_rdma_size_hint", &deep_ep::get_low_latency_rdma_size_hint);

    py::class_<deep_ep::EventHandle>(m, "EventHandle")
        .def(py::init<>())
        .def("current_stream_wait", &deep_ep::EventHandle::current_stream_wait);

    py::class_<deep_ep::Buffer>(m, "Buffer")
        .def(py::init<int, int, int64_t, int64_t, bool, std::string>())
        // MI355X window bootstrap
        .def("get_local_device_id", &deep_ep::Buffer::get_local_device_id)
        .def("get_local_ipc_handle", [](const deep_ep::Buffer &b) { return py::bytes(b.get_local_ipc_handle()); })
        .def("get_local_window_ptr", &deep_ep::Buffer::get_local_window_ptr)
        .def("get_window_bytes", &deep_ep::Buffer::get_window_bytes)
        .def("sync", &deep_ep::Buffer::sync, py::arg("handles"), py::arg("local_ptrs"))
        // reference surface
        .def("is_available", &deep_ep::Buffer::is_available)
        .def("get_num_rdma_ranks", &deep_ep::Buffer::get_num_rdma_ranks)
        .def("get_rdma_rank", &deep_ep::Buffer::get_rdma_rank)
        .def("get_dispatch_layout", &deep_ep::Buffer::get_dispatch_layout)
        .def("get_notify_send_data", &deep_ep::Buffer::get_notify_send_data)
        .def("clean_low_latency_buffer", &deep_ep::Buffer::clean_low_latency_buffer)
        .def("intranode_dispatch", &deep_ep::Buffer::intranode_dispatch)
        .def("notify_verify", &deep_ep::Buffer::notify_verify)
        .def("intranode_combine", &deep_ep::Buffer::intranode_combine)
        .def("internode_dispatch", [](deep_ep::Buffer &b, py::args, py::kwargs) { b.internode_unsupported(); })
        .def("internode_combine", [](deep_ep::Buffer &b, py::args, py::kwargs) { b.internode_unsupported(); })
        .def("low_latency_dispatch", &deep_ep::Buffer::low_latency_dispatch)
        .def("low_latency_combine", &deep_ep::Buffer::low_latency_combine)
        .def("fused_deep_moe", &deep_ep::Buffer::fused_deep_moe, py::arg("x"), py::arg("expert_ids"),
             py::arg("gmm1_permuted_weight"), py::arg("gmm1_permuted_weight_scale"), py::arg("gmm2_weight"),
             py::arg("gmm2_weight_scale"), py::arg("expert_scales_optional"), py::arg("num_max_dispatch_tokens_per_rank"),
             py::arg("num_experts"), py::arg("quant_mode"), py::arg("profile_enable") = false)
        .def("begin_profile", &deep_ep::Buffer::begin_profile, py::arg("num_profile_skip_launches"),
             py::arg("num_profile_active_launches"), py::arg("profile_trace_dir") = "")
        .def("end_profile", &deep_ep::Buffer::end_profile)
        .def("get_profile_summary", &deep_ep::Buffer::get_profile_summary)
        .def("dispatch_ffn_combine", &deep_ep::Buffer::dispatch_ffn_combine)
        // alltoall-strategy kernel entry points
        .def("a2a_dispatch_stage", &deep_ep::Buffer::a2a_dispatch_stage)
        .def("a2a_dispatch_tables", &deep_ep::Buffer::a2a_dispatch_tables)
        .def("a2a_dispatch_unpack", &deep_ep::Buffer::a2a_dispatch_unpack)
        .def("a2a_combine_pack", &deep_ep::Buffer::a2a_combine_pack)
        .def("a2a_combine_prepare", &deep_ep::Buffer::a2a_combine_prepare)
        .def("a2a_combine_reduce", &deep_ep::Buffer::a2a_combine_reduce);
}
