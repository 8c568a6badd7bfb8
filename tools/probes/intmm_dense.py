import torch, time
torch.cuda.set_device(0)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (M,K,N) in ((32768,7168,4096),(32768,2048,7168),(1024,7168,4096),(4096,7168,4096)):
    a=torch.randint(-8,8,(M,K),dtype=torch.int8,device='cuda'); w=torch.randint(-8,8,(N,K),dtype=torch.int8,device='cuda')
    ms=t(lambda: torch._int_mm(a, w.t()))
    print(f"_int_mm {M}x{K}x{N}: {ms*1e3:.0f} us  {2*M*K*N/ms/1e9:.0f} TOPS")
    ab=a.to(torch.bfloat16); wb=w.to(torch.bfloat16)
    ms=t(lambda: ab @ wb.t())
    print(f"bf16 mm {M}x{K}x{N}: {ms*1e3:.0f} us  {2*M*K*N/ms/1e9:.0f} TFLOPS")
