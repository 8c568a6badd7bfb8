#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 passes of the default bench command, then the tracked summaries.
#   bash tools/collect_profiles.sh <round>
# Pass 1: --kernel-trace --stats (per-kernel durations).  Passes 2 and 3: --pmc FETCH_SIZE / --pmc WRITE_SIZE, each in its own run
# with --kernel-trace only (never combined with the hip/hsa/memory-copy trace domains).  tools/summarize_prof.py FAILS when one of
# the kernels bench.py prices (PMC_KERNEL_NAMES + the MLA kernels) is missing from the counters, so profiles/r<round>_pmc_traffic.json
# cannot silently go stale against renamed kernels.
set -u
RND=${1:-02}
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
python tools/summarize_prof.py "$RND" "$S" "$F" "$W" || { echo "summarize_prof failed: profiles NOT refreshed" >&2; exit 1; }
grep "^{" "$OUT/bench_under_rocprof.out" | tail -1 > profiles/r${RND}_bench_n1_under_rocprof.json
python bench.py 2>/dev/null | grep "^{" | tail -1 > profiles/r${RND}_bench_n1.json
# secondary measurements (elementwise primitives, GQA decode, mla_preprocess)
python tools/bench_extra.py 2>/dev/null | grep "^{" | tail -1 > profiles/r${RND}_extra_bench_n1.json
mkdir -p gpurun_out/profiles_r$RND && cp profiles/r${RND}_* gpurun_out/profiles_r$RND/
head -14 profiles/r${RND}_kernel_stats.csv
