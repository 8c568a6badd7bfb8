"""Scratch: which torch GPU matmul is exact for int8-valued operands (checker infrastructure)?"""
import torch
g = torch.Generator().manual_seed(0)
for M in (8, 16, 17, 32, 64, 200):
    a = torch.randint(-127, 128, (M, 7168), generator=g, dtype=torch.int32)
    w = torch.randint(-16, 16, (4096, 7168), generator=g, dtype=torch.int32)
    want = (a.double() @ w.double().t())
    ad, wd = a.cuda(), w.cuda()
    f64 = (ad.double() @ wd.double().t()).cpu()
    f32 = (ad.float() @ wd.float().t()).double().cpu()
    Mp = max(32, (M + 7) // 8 * 8)
    ap = torch.zeros((Mp, 7168), dtype=torch.int8, device="cuda"); ap[:M] = ad.to(torch.int8)
    i8 = torch._int_mm(ap, wd.to(torch.int8).t())[:M].double().cpu()
    print(M, "f64 max err", (f64 - want).abs().max().item(), "f32", (f32 - want).abs().max().item(), "int_mm", (i8 - want).abs().max().item())
