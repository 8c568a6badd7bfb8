import torch
a = torch.randint(-127, 127, (32, 7168), dtype=torch.int8, device="cuda")
w = torch.randint(-127, 127, (2112, 7168), dtype=torch.int8, device="cuda")
try:
    c = torch._int_mm(a, w.t())
    ref = (a.cpu().int() @ w.cpu().int().t())
    print("int_mm ok", c.dtype, c.shape, torch.equal(c.cpu(), ref))
    import time
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(50): c = torch._int_mm(a, w.t())
    torch.cuda.synchronize(); print("us", (time.perf_counter()-t)/50*1e6)
except Exception as e:
    print("int_mm failed", e)
