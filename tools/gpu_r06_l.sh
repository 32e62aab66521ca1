#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "512 512" "512 256" "256 256" "256 384" "512 128"; do
  set -- $cfg
  echo "== split MINTILES=$1 MINTILES2=$2"; SPH3D_SPLIT_MINTILES=$1 SPH3D_SPLIT_MINTILES2=$2 timeout 300 python tools/exp_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/| blas.*//'
  v=$(SPH3D_SPLIT_MINTILES=$1 SPH3D_SPLIT_MINTILES2=$2 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "headline: $v"
done | tee $OUT/r06_exp_gemm_split_tiles2.log
