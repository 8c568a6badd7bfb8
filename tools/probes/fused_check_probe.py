"""Scratch: HIP fused_deep_moe vs the sampled per-token checker (GPU and CPU) vs the NumPy oracle on one mid-size case."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "sgl-kernel-npu_amd", "python")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import fused_f64 as F
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_device(0)
import deep_ep
from oracle import ep as O
from oracle.bf16 import torch_to_bits, bf16_bits_to_f32
T, H, I, K, L = int(sys.argv[1]), 7168, 2048, 8, int(sys.argv[2])
E = L
buf = deep_ep.Buffer(dist.group.WORLD, low_latency_mode=True)
w13, w2, s13, s2 = F.fused_weights(900, L, H, I)
perm = F.fusion_perm(2 * I)
w13_p, s13_p = w13[:, perm, :].contiguous(), s13[:, perm].contiguous()
g = torch.Generator(device="cuda").manual_seed(1900)
x = torch.randn((T, H), generator=g, device="cuda").to(torch.bfloat16)
idx = torch.topk(torch.rand((T, E), generator=g, device="cuda"), K, dim=-1)[1]
w = torch.rand((T, K), generator=g, device="cuda")
for _ in range(2):
    out, _ = buf.fused_deep_moe(x, idx, w, w13_p, s13_p, w2, s2, T, E)
torch.cuda.synchronize()
sel = torch.arange(0, T, max(1, T // 64), device="cuda")[:64]
ref_gpu = F.sampled_reference(x, idx, w, lambda r: (w13, w2, s13, s2), L, sel)
print("HIP vs GPU checker:", F.diffs(out[sel], ref_gpu))
cw = (w13.cpu(), w2.cpu(), s13.cpu(), s2.cpu())
ref_cpu = F.sampled_reference(x.cpu(), idx.cpu(), w.cpu(), lambda r: cw, L, sel[:16].cpu())
print("GPU checker vs CPU checker:", F.diffs(ref_gpu[:16].cpu(), ref_cpu), "HIP vs CPU checker:", F.diffs(out[sel[:16]].cpu(), ref_cpu))
if T <= 1024:
    want = O.fused_deep_moe([torch_to_bits(x)], [idx.cpu().numpy()], [w.cpu().numpy()], [cw[0].numpy()], [cw[2].numpy()], [cw[1].numpy()], [cw[3].numpy()], T, E)[0]
    wt = torch.from_numpy(bf16_bits_to_f32(want))
    print("HIP vs oracle:", F.diffs(out.cpu(), wt), "GPU checker vs oracle:", F.diffs(ref_gpu.cpu(), wt[sel.cpu()]))
per = ((out[sel].double() - ref_gpu).abs() / ref_gpu.abs().clamp_min(1e-2)).mean(dim=1)
print("per-sample avg diff:", [f"{v:.1e}" for v in per.tolist()])
# first quantisation: checker vs oracle for the sampled tokens
xs_ = x[sel].float(); amax = xs_.abs().amax(dim=1, keepdim=True); s_ = 127.0 / amax
q_chk = torch.round(xs_ * s_).to(torch.int8).cpu().numpy()
q_or, sc_or = O.quant_int8_rows(torch_to_bits(x[sel]), None)
print("x quant equal:", np.array_equal(q_chk, q_or), "scale equal:", np.array_equal((1.0 / s_).cpu().numpy().reshape(-1), sc_or))
# one expert of one bad token, stage by stage against the oracle's functions
bad = int(per.argmax()); t = int(sel[bad]); e = int(idx[t, 0])
a = q_or[bad:bad + 1]; asc = sc_or[bad:bad + 1]
v_or = O.moe_gemm1_swiglu(a, asc, w13[e].cpu().numpy(), s13[e].cpu().numpy())
c = (torch.from_numpy(a).double().cuda() @ w13[e].double().t()).float()
d = (c * s13[e][None, :]) * torch.from_numpy(asc).cuda()[:, None]
gate, up = d[:, :I], d[:, I:]
v_chk = (up * (gate / (1 + torch.exp(-gate)))).cpu().numpy()
print("token", t, "expert", e, "v equal:", np.array_equal(v_or, v_chk), "max rel dv", np.abs(v_or - v_chk).max() / np.abs(v_or).max(),
      "n differing", int((v_or != v_chk).sum()))
q2_or, s2_or = O.moe_rowquant(v_or); q2_c, s2_c = O.moe_rowquant(v_chk)
print("q2 flips:", int((q2_or != q2_c).sum()), "of", q2_or.size)
sc_chk = (1.0 / s_).cpu().numpy().reshape(-1)
for j in range(len(sel)):
    nd = int((q_chk[j] != q_or[j]).sum())
    if nd or sc_chk[j] != sc_or[j]:
        print("sample", j, "q diffs", nd, "scale chk/or", float(sc_chk[j]).hex(), float(sc_or[j]).hex(), "amax", float(amax[j]).hex(), "s gpu", float(s_[j]).hex(),
              "s np", float(np.float32(127.0) / np.float32(float(amax[j]))).hex(), "max|q diff|", int(np.abs(q_chk[j].astype(int) - q_or[j].astype(int)).max()))
