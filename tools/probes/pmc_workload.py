"""Workload of tools/collect_counters.sh: BASELINE C4 (MLA decode, both wide kernel forms) and C5 (fused_deep_moe, 4096 tokens, 32 local
experts) at full size, a fixed number of launches each, nothing else heavy -- so that per-kernel PMC medians are per-config."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "sgl-kernel-npu_amd", "python")):
    sys.path.insert(0, p)
import ctypes, torch, torch.distributed as dist
from sgl_kernel_npu.bench_hooks import _mla_inputs
import sgl_kernel_npu
q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64)
out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device="cuda")
lib = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_sgl_kernels.so"), mode=ctypes.RTLD_GLOBAL)
ONLY_FUSED = os.environ.get("PMC_ONLY_FUSED") == "1"          # tools/probes/gemm_traffic_ab.sh: the C5 leg alone
for waves in (() if ONLY_FUSED else (4, 8, 9)):      # four waves | eight waves, block-id ring | eight waves, scalar block ids (the default)
    lib.mi_mla_decode_select_wide(waves)
    for _ in range(40):
        torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, 0)
    torch.cuda.synchronize()
lib.mi_mla_decode_select_wide(0)
# the reference test's GQA shape at serving size (128 q heads on one kv head, 288 / 256, V = a column prefix of K): gqa_decode_wide_kernel<.., true, 64>
from sgl_kernel_npu.attention.decode_attention import decode_gqa
gk = torch.randn((128 * 64, 64, 1, 288), device="cuda").to(torch.bfloat16)
gq = torch.randn((128, 128, 288), device="cuda").to(torch.bfloat16)
gbt = torch.randperm(128 * 64, device="cuda").to(torch.int32).reshape(128, 64)
gl = torch.full((128,), 4096, dtype=torch.int32, device="cuda")
go = torch.empty((128, 128, 256), dtype=torch.bfloat16, device="cuda")
for _ in range(0 if ONLY_FUSED else 40):
    decode_gqa(gq, gk, gk[..., :256], go, gl, 288 ** -0.5, 64, gbt)
torch.cuda.synchronize()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29581")
dist.init_process_group("gloo", rank=0, world_size=1)
import deep_ep, fused_f64 as F
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
T, H, I, K, Lx = 4096, 7168, 2048, 8, 32
w13, w2, s13, s2 = F.fused_weights(99, Lx, H, I)
g = torch.Generator(device="cuda").manual_seed(199)
x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, Lx), generator=g, device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), generator=g, device="cuda")
for _ in range(12):
    buf.fused_deep_moe(x, idx, w, w13, s13, w2, s2, T, Lx)
torch.cuda.synchronize()
