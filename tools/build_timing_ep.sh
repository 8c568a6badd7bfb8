#!/bin/bash
# Instrumented / experimental copy of libmi_ep.so: tools/build_timing_ep.sh <suffix> -DX=.. -> sgl-kernel-npu_amd/lib/timing_ep/libmi_ep_<suffix>.so
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SFX="$1"; shift
OUT="$ROOT/sgl-kernel-npu_amd/lib/timing_ep"
mkdir -p "$OUT/obj_$SFX"
pids=()
for f in "$ROOT"/sgl-kernel-npu_amd/csrc/ep/*.hip; do
  b=$(basename "$f" .hip)
  extra=$(head -1 "$f" | sed -n 's#^// hipcc-flags:##p')
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-strict-aliasing -I"$ROOT/include" -I"$ROOT/sgl-kernel-npu_amd/csrc/ep" \
        -I"$ROOT/sgl-kernel-npu_amd/csrc" $extra "$@" -c "$f" -o "$OUT/obj_$SFX/$b.o" 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC "$OUT/obj_$SFX"/*.o -o "$OUT/libmi_ep_$SFX.so"
echo "$OUT/libmi_ep_$SFX.so"
