import torch, sys
sys.path.insert(0, "sgl-kernel-npu_amd/python")
from sgl_kernel_npu.activation.swiglu_oai_quant import swiglu_oai_quant
x = torch.randn(16384, 5760, device="cuda").to(torch.bfloat16)
for nq in (True, False):
    for _ in range(3): swiglu_oai_quant(x, 1.702, 7.0, nq)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): swiglu_oai_quant(x, 1.702, 7.0, nq)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(nq, us, "us", 16384 * (5760 * 2 + 2880 * (1 if nq else 2)) / us / 1e3, "GB/s")
