#!/bin/bash
# HIP API statistics of a short bench run: which runtime calls does the steady-state step make, and how long do they block?
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/hipt -o h -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/hiptrace.log 2>&1
echo "rc=$?"
f=$(find /tmp/hipt -name "*hip_api_stats.csv" | head -1)
head -40 $f | cut -c1-160
cp $f $GRAFT_REPO_ROOT/gpurun_out/hip_api_stats.csv
