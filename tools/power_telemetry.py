"""Socket power, shader / memory clock and throttle state sampled at >= 10 Hz while ONE kernel loops for several seconds.

  python tools/power_telemetry.py <workload> [seconds] > sample.json
  workload: mla_c4 | mla_ragged | gemm1 | gemm2 | gqa | idle
  (kernel ablations: LD_PRELOAD a tools/build_timing.sh build of libmi_sgl_kernels, e.g. -DMLA8S_NO_DMA = the tile loop without its KV fill,
   -DMLA8S_NO_QK -DMLA8S_NO_PV = the fill + softmax alone; tools/probes/mla_power.sh runs the three MLA legs and writes profiles/r06_mla_power.json)

The sampler is a thread of this process that reads the amdgpu hwmon files of the busiest GPU (power1_input = socket power in microwatts,
freq1_input = sclk, freq2_input = mclk, power1_cap) every 50 ms and calls `amd-smi metric --json` every ~0.5 s for the throttle / clock-lock status and
the per-XCD clocks.  The workload thread queues launches in chunks and never lets the GPU idle.  Output: one JSON object with the samples'
summary (median / p10 / p90 over the steady part: the first second is dropped) and the kernel's average duration over the same window."""
import ctypes
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sgl-kernel-npu_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402


def hwmon_dirs():
    return [d for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if os.path.exists(os.path.join(d, "power1_input"))]


def rd(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def make_workload(name):
    if name == "idle":
        return lambda: time.sleep(0.01), 1
    if name in ("mla_c4", "mla_ragged"):
        from sgl_kernel_npu.bench_hooks import _mla_inputs
        import sgl_kernel_npu  # noqa: F401
        q, kn, kr, bt, lens = _mla_inputs(128, 128, 4096, 64, ragged=(name == "mla_ragged"))
        out = torch.empty((128, 128, 512), dtype=torch.bfloat16, device="cuda")
        return (lambda: torch.ops.npu.decode_mla(q, kn, kr, out, lens, 576 ** -0.5, 64, bt, 0)), 50
    if name == "gqa":
        from sgl_kernel_npu.attention.decode_attention import decode_gqa
        Bq, Hq, Hkv, D, Dv, Sq, page = 128, 128, 1, 288, 256, 4096, 64          # the reference test's shape
        g = torch.Generator(device="cuda").manual_seed(7)
        nb = Bq * Sq // page
        q = torch.randn((Bq, Hq, D), generator=g, device="cuda").to(torch.bfloat16)
        kc = torch.randn((nb, page, Hkv, D), generator=g, device="cuda").to(torch.bfloat16)
        vc = kc[..., :Dv]
        bt = torch.randperm(nb, device="cuda").to(torch.int32).reshape(Bq, Sq // page)
        lens = torch.full((Bq,), Sq, dtype=torch.int32, device="cuda")
        o = torch.empty((Bq, Hq, Dv), device="cuda", dtype=torch.bfloat16)
        return (lambda: decode_gqa(q, kc, vc, o, lens, D ** -0.5, page, bt)), 50
    if name in ("gemm1", "gemm2"):
        from capi import ptr, stream_ptr
        L = ctypes.CDLL(os.path.join(ROOT, "sgl-kernel-npu_amd", "lib", "libmi_ep.so"))
        E, H, I2, M = 32, 7168, 4096, 32768
        c_vp = ctypes.c_void_p
        sig = [c_vp] * 5 + [ctypes.c_int] * 5 + [c_vp, ctypes.c_int, c_vp]
        L.mi_ep_moe_gemm1_swiglu.argtypes = sig
        L.mi_ep_moe_gemm2.argtypes = sig
        K, N = (H, I2) if name == "gemm1" else (I2 // 2, H)
        gen = torch.Generator().manual_seed(3)
        cnt = torch.bincount(torch.multinomial(torch.ones(E), M, replacement=True, generator=gen), minlength=E)
        cum = torch.cumsum(cnt, 0).to(torch.int32).cuda().contiguous()
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device="cuda")
        w = torch.randint(-127, 128, (E, N, K), dtype=torch.int8, device="cuda")
        asc, ws = torch.rand(M, device="cuda"), torch.rand((E, N), device="cuda")
        if name == "gemm1":
            out = torch.zeros((M, N // 2), dtype=torch.float32, device="cuda")
            return (lambda: L.mi_ep_moe_gemm1_swiglu(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, K, N, ptr(out), 0, stream_ptr())), 10
        out = torch.zeros((M, N), dtype=torch.bfloat16, device="cuda")
        return (lambda: L.mi_ep_moe_gemm2(ptr(a), ptr(asc), ptr(w), ptr(ws), ptr(cum), 1, E, M, K, N, ptr(out), 0, stream_ptr())), 10
    raise SystemExit(f"unknown workload {name}")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "mla_c4"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    f, chunk = make_workload(name)
    dirs = hwmon_dirs()
    samples, smi = [], []
    stop = threading.Event()
    t_start = time.time()

    def sampler():
        k = 0
        while not stop.is_set():
            t = time.time() - t_start
            row = {"t": round(t, 3)}
            for d in dirs:
                card = d.split("/")[4]
                row[card] = {"power_uW": rd(os.path.join(d, "power1_input")), "sclk_Hz": rd(os.path.join(d, "freq1_input")),
                             "mclk_Hz": rd(os.path.join(d, "freq2_input"))}
            samples.append(row)
            if k % 10 == 0:
                try:
                    r = subprocess.run(["amd-smi", "metric", "--json"], capture_output=True, text=True, timeout=5)
                    j = json.loads(r.stdout)
                    g = (j.get("gpu_data") or j)[0] if isinstance(j, (dict, list)) else {}
                    smi.append({"t": round(t, 3), "power": g.get("power"), "clock_gfx": (g.get("clock") or {}).get("gfx_0"),
                                "clock_mem": (g.get("clock") or {}).get("mem_0"), "throttle": g.get("throttle"),
                                "temperature": g.get("temperature")})
                except Exception as e:  # noqa: BLE001
                    smi.append({"t": round(t, 3), "error": str(e)[:200]})
            k += 1
            time.sleep(0.05)

    for _ in range(3):
        f()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    calls, gpu_ms = 0, 0.0
    while time.time() - t_start < seconds:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(chunk):
            f()
        b.record()
        torch.cuda.synchronize()
        if time.time() - t_start > 1.0:
            calls += chunk
            gpu_ms += a.elapsed_time(b)
    stop.set()
    th.join()
    # the GPU this process ran on = the card whose power moved most
    steady = [s for s in samples if s["t"] > 1.0]
    cards = [k for k in steady[0] if k != "t"] if steady else []
    def med(v):
        v = sorted(x for x in v if x is not None)
        return v[len(v) // 2] if v else None
    def pct(v, p):
        v = sorted(x for x in v if x is not None)
        return v[min(len(v) - 1, int(len(v) * p))] if v else None
    busiest = None
    try:      # the card of THIS process's GPU by PCI address (other GPUs of the host may be busy with other jobs)
        pr = torch.cuda.get_device_properties(0)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        for c in cards:
            if addr in os.path.realpath(f"/sys/class/drm/{c}/device"):
                busiest = c
    except Exception:  # noqa: BLE001
        pass
    out_match = "pci" if busiest else "busiest"
    if busiest is None:
        busiest = max(cards, key=lambda c: med([s[c]["power_uW"] for s in steady]) or 0) if cards else None
    out = {"workload": name, "ld_preload": os.environ.get("LD_PRELOAD", ""), "seconds": seconds, "samples": len(steady),
           "sample_hz": round(len(steady) / max(1e-9, seconds - 1.0), 1),
           "kernel_avg_us": round(gpu_ms * 1e3 / calls, 2) if calls else None, "card": busiest, "card_matched_by": out_match}
    if busiest:
        d = [x for x in dirs if x.split("/")[4] == busiest][0]
        out["power_cap_W"] = (rd(os.path.join(d, "power1_cap")) or 0) / 1e6
        pw = [s[busiest]["power_uW"] for s in steady]
        sc = [s[busiest]["sclk_Hz"] for s in steady]
        mc = [s[busiest]["mclk_Hz"] for s in steady]
        out["socket_power_W"] = {"p10": (pct(pw, 0.1) or 0) / 1e6, "median": (med(pw) or 0) / 1e6, "p90": (pct(pw, 0.9) or 0) / 1e6, "max": (max(x for x in pw if x is not None) if pw else 0) / 1e6}
        out["sclk_MHz"] = {"p10": (pct(sc, 0.1) or 0) / 1e6, "median": (med(sc) or 0) / 1e6, "p90": (pct(sc, 0.9) or 0) / 1e6}
        out["mclk_MHz"] = {"median": (med(mc) or 0) / 1e6}
        out["other_cards_median_W"] = {c: (med([s[c]["power_uW"] for s in steady]) or 0) / 1e6 for c in cards if c != busiest}
    out["amd_smi"] = smi[1:6]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
