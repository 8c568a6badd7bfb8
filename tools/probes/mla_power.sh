#!/bin/bash
# profiles/r06_mla_power.json: socket power / sclk / mclk / throttle state at >= 10 Hz across 6 s loops of (i) the full C4 MLA decode kernel,
# (ii) its KV fill + softmax alone (no QK^T, no P.V MFMAs), (iii) the tile loop without the fill, plus the grouped INT8 GEMMs, the wide GQA kernel and
# an idle leg, with the board's power cap beside them.  Run on the GPU box: bash tools/probes/mla_power.sh > gpurun_out/r06_mla_power.json
set -e
cd "$(dirname "$0")/../.."
T=sgl-kernel-npu_amd/lib/timing
[ -f $T/libmi_sgl_kernels_nofill.so ] || bash tools/build_timing.sh nofill -DMLA8S_NO_DMA >/dev/null
[ -f $T/libmi_sgl_kernels_fillonly.so ] || bash tools/build_timing.sh fillonly -DMLA8S_NO_QK -DMLA8S_NO_PV >/dev/null
echo "{"
echo "\"idle\": $(python tools/power_telemetry.py idle 3 2>/dev/null | tail -1),"
echo "\"mla_c4_full\": $(python tools/power_telemetry.py mla_c4 6 2>/dev/null | tail -1),"
echo "\"mla_c4_fill_only\": $(LD_PRELOAD=$T/libmi_sgl_kernels_fillonly.so python tools/power_telemetry.py mla_c4 6 2>/dev/null | tail -1),"
echo "\"mla_c4_no_fill\": $(LD_PRELOAD=$T/libmi_sgl_kernels_nofill.so python tools/power_telemetry.py mla_c4 6 2>/dev/null | tail -1),"
echo "\"mla_ragged_full\": $(python tools/power_telemetry.py mla_ragged 6 2>/dev/null | tail -1),"
echo "\"gqa_wide_288_256\": $(python tools/power_telemetry.py gqa 6 2>/dev/null | tail -1),"
echo "\"moe_gemm1_c5\": $(python tools/power_telemetry.py gemm1 6 2>/dev/null | tail -1),"
echo "\"moe_gemm2_c5\": $(python tools/power_telemetry.py gemm2 6 2>/dev/null | tail -1)"
echo "}"
