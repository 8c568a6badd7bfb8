#!/bin/bash
# Runs ON the GPU box (through gpurun): SQ / GRBM counter passes (rocprofv3 --pmc with --kernel-trace only, one small group per pass)
# over tools/probes/pmc_workload.py, summarised into profiles/r<round>_pmc_mfma_lds.json: MFMA-busy, LDS bank conflicts, issue stalls
# for the MLA decode kernels (C4) and the grouped INT8 GEMMs (C5).
#   bash tools/collect_counters.sh <round>
set -u
RND=${1:-03}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_r$RND
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
# (the last two passes: fabric-side traffic of the same launches -- the kernels bench.py does not run, e.g. the wide GQA kernel, get their FETCH / WRITE here)
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$REPO/tools/probes/pmc_workload.py" > /dev/null 2>&1
done
cd "$REPO"
python - "$OUT" "$RND" <<'PY'
import csv, glob, json, os, re, sys, collections
out, rnd = sys.argv[1], sys.argv[2]
want = ("mla_decode_wide_kernel", "mla_decode_wide8_kernel", "mla_decode_wide8s_kernel", "mla_merge_kernel", "gqa_decode_wide_kernel", "grouped_gemm_i8_kernel", "rowquant_kernel")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        m = re.search(r"([A-Za-z0-9_]+)(<[^(]*>)?\(", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:50]
        if name.startswith(want):
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(out, "p1", "*", "*kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        m = re.search(r"([A-Za-z0-9_]+)(<[^(]*>)?\(", r["Kernel_Name"])
        name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:50]
        if name.startswith(want):
            dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
med = lambda v: sorted(v)[len(v) // 2] if v else None
res = {}
for k, cs in sorted(agg.items()):
    c = {n: med(v) for n, v in cs.items()}
    d = {"launches": max(len(v) for v in cs.values()), "duration_us_under_pmc": (med(dur[k]) or 0) / 1e3, "counters_median": c}
    # waves that share a SIMD while the kernel runs (one workgroup per CU in all of them): 256 threads -> 1, 512 -> 2, 1024 -> 4
    wps = 2 if ("wide8" in k or k.startswith("gqa_decode_wide")) else (4 if k.startswith("grouped_gemm") else 1)
    der = {}
    if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("SQ_WAVE_CYCLES"):
        per_wave = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"])
        der["mfma_busy_cycles_per_wave_resident_cycle"] = per_wave
        der["waves_per_simd"] = wps
        der["mfma_busy_fraction_of_simd_time"] = per_wave * wps
    if c.get("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_over_lds_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
        der["hbm_side_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0      # gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md)
    if c.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c.get(n) is not None:
                der[n.lower() + "_over_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    d["derived"] = der
    res[k] = d
note = ("rocprofv3 --pmc passes (tools/collect_counters.sh) over tools/probes/pmc_workload.py: BASELINE C4 MLA decode (bs 128 x 128 heads x 4096 "
        "keys, 2 KV splits; four-wave and eight-wave wide kernels) and C5 fused_deep_moe (4096 tokens, 32 local experts: grouped_gemm_i8 "
        "<0,..> = GEMM1 + SwiGLU, <2,..> = GEMM2 + combine push).  Medians over the launches.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are "
        "quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md); mfma_busy_fraction_of_simd_time = MFMA busy cycles per "
        "wave-resident cycle x the waves that share a SIMD.")
json.dump({"note": note, "kernels": res}, open(os.path.join("profiles", f"r{rnd}_pmc_mfma_lds.json"), "w"), indent=1)
for k, d in res.items():
    print(k, round(d["duration_us_under_pmc"], 1), {a: round(b, 3) for a, b in d["derived"].items()})
PY
mkdir -p gpurun_out/profiles_r$RND && cp profiles/r${RND}_pmc_mfma_lds.json gpurun_out/profiles_r$RND/
