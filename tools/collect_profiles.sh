#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 passes of the default bench command, then the tracked summaries.
#   bash tools/collect_profiles.sh <round>
set -u
RND=${1:-01}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_r$RND
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$REPO/bench.py" > "$OUT/bench_under_rocprof.out" 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python "$REPO/bench.py" --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python "$REPO/bench.py" --no-cpu-baseline > /dev/null 2>&1
cd "$REPO"
S=$(dirname "$(ls $OUT/stats/*/*kernel_stats.csv | head -1)")
F=$(dirname "$(ls $OUT/fetch/*/*counter_collection.csv | head -1)")
W=$(dirname "$(ls $OUT/write/*/*counter_collection.csv | head -1)")
python tools/summarize_prof.py "$RND" "$S" "$F" "$W"
grep "^{" "$OUT/bench_under_rocprof.out" | tail -1 > profiles/r${RND}_bench_n1_under_rocprof.json
cp profiles/r${RND}_* gpurun_out/ 2>/dev/null
python bench.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r${RND}_bench_n1.json
head -12 profiles/r${RND}_kernel_stats.csv
# secondary configs (LL latency, fused_deep_moe, primitives, GQA, mla_preprocess)
python tools/bench_extra.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r${RND}_extra_bench_n1.json
# (MLA wide-kernel PMC counters: bash tools/pmc_mla.sh, medians go into profiles/r${RND}_pmc_mla_decode.json)
