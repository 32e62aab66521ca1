#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (no PMC), results as CSV under gpurun_out/prof_<tag>
TAG=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
echo "prof rc=$?"
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_$TAG; rm -f gpurun_out/prof_$TAG/*kernel_trace.csv.bak
# keep the merged output small: drop the per-dispatch trace if it is huge
find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -size +20M -delete
