#!/bin/bash
# DEEPEP_ROCTX=1: do the kernel chains show up as roctx ranges?  (rocprofv3 --marker-trace --kernel-trace, no counters)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/roctx_chk
DEEPEP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/roctx_chk -- python "$GRAFT_REPO_ROOT/bench.py" --no-mla --no-cpu-baseline --no-extra --steps 3 --warmup 1 2>&1 | tail -5
f=$(find /tmp/roctx_chk -name "*marker_api_trace.csv" | head -1)
find /tmp/roctx_chk -type f | head; echo "marker file: $f"; [ -n "$f" ] && cut -d, -f1-3 "$f" | sort | uniq -c | sort -rn | head -12
