import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
import torch
import sgl_kernel_npu
from oracle import kernels as OK
torch.manual_seed(2)
for (B, Hq, S, page) in [(3, 32, 129, 16), (1, 256, 150, 64), (2, 16, 200, 64)]:
    maxp = (S + page - 1) // page; nb = B * maxp + 3
    dtype = torch.bfloat16
    q = torch.randn((B, Hq, 576)).to(dtype); kn = torch.randn((nb, page, 1, 512)).to(dtype); kr = torch.randn((nb, page, 1, 64)).to(dtype)
    bt = torch.randperm(nb)[:B * maxp].to(torch.int32).reshape(B, maxp)
    lens = torch.tensor([max(1, S - 37 * i) for i in range(B)], dtype=torch.int32)
    sm = 576 ** -0.5
    want = OK.decode_mla(q, kn, kr, lens, bt, sm).float()
    out = torch.empty((B, Hq, 512), dtype=dtype, device="cuda")
    torch.ops.npu.decode_mla(q.cuda(), kn.cuda(), kr.cuda(), out, lens.cuda(), sm, page, bt.cuda(), 1)
    got = out.cpu().float()
    # exact fp32/fp64
    ex = torch.zeros_like(want)
    for b in range(B):
        L = int(lens[b]); idx = bt[b, :(L + page - 1) // page].long()
        K = torch.cat([kn[idx].reshape(-1, 512), kr[idx].reshape(-1, 64)], 1)[:L].double()
        p = torch.softmax((q[b].double() @ K.T) * sm, -1)
        ex[b] = (p @ K[:, :512]).float()
    d = (got - want).abs(); viol = d - (1e-3 + 2 ** -7 * want.abs())
    i = viol.argmax()
    print(B, Hq, S, "max|got-oracle|", d.max().item(), "max viol", viol.max().item(), "at got", got.flatten()[i].item(), "oracle", want.flatten()[i].item(), "exact", ex.flatten()[i].item(),
          "| max|got-exact|", (got - ex).abs().max().item(), "max|oracle-exact|", (want - ex).abs().max().item())
