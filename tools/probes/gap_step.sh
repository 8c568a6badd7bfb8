cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gp; rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/bench.py --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/probes/gap_analysis.py $f
python $R/tools/trace_step.py $f | tail -12
