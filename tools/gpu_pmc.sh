#!/bin/bash
# usage: gpu_pmc.sh <tag> <script.py> "<counters>"   (separate PMC pass, kernel-trace only)
TAG=$1; SCRIPT=$2; CNT=$3
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && timeout -k 5 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/$SCRIPT > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1
echo rc=$?
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
