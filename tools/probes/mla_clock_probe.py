"""Scratch: does MLA decode at C4 run clock/power-limited?  Per-batch time series of back-to-back calls + rocm-smi samples taken while a
long batch is queued."""
import os, subprocess, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "sgl-kernel-npu_amd", "python"))
from sgl_kernel_npu.bench_hooks import _mla_inputs
import sgl_kernel_npu
q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64)
out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device="cuda")
f = lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, 0)
f(); torch.cuda.synchronize(); time.sleep(2.0)
series = []
for batch in range(40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(25): f()
    b.record(); torch.cuda.synchronize()
    series.append(a.elapsed_time(b) / 25 * 1e3)
print("us per call, batches of 25 from idle:", " ".join(f"{v:.0f}" for v in series), flush=True)
for _ in range(12000): f()          # ~2 s of queued work
time.sleep(1.0)
r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True)
print("\n".join(l for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)", "fclk"))), flush=True)
torch.cuda.synchronize()
