#!/bin/bash
# round 6, call B: whole GPU test suite on the new transposed-graph build (single-pass scan, fill+order in one launch), then the
# headline A/B against the commit before it (build_exp/base), and the launch count of a step
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
bash tools/gpu_ab_tree.sh 3 80 | tee $OUT/r06_ab_tg_merge.log
