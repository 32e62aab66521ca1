#!/bin/bash
# kernel trace of a short bench run + timeline analysis (tools/analyze_trace.py); the trace stays on the GPU box
TAG=${1:-t}
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.log 2>&1
echo "rc=$?"
f=$(find /tmp/trace_$TAG -name "*kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/analyze_trace.py $f | tee $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.txt
