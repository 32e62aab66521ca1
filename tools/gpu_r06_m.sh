#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -3
for mk in 100000 512 1024 256; do
  echo "== split BK32_MINK=$mk"; SPH3D_SPLIT_BK32_MINK=$mk timeout 300 python tools/exp_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/| blas.*//'
  v=$(SPH3D_SPLIT_BK32_MINK=$mk timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "headline: $v"
done | tee $OUT/r06_exp_gemm_split_bk32.log
