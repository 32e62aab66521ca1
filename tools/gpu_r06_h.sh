#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -3
for mt in 512 256 128; do
  echo "== split, SPH3D_SPLIT_MINTILES=$mt"; SPH3D_SPLIT_MINTILES=$mt timeout 300 python tools/exp_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/| blas.*//'
done | tee $OUT/r06_exp_gemm_split_tiles.log
for i in 1 2; do
  for mt in 512 256; do
  a=$(SPH3D_SPLIT_MINTILES=$mt timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $i: split mintiles $mt: $a"
  done
done | tee -a $OUT/r06_exp_gemm_split_tiles.log
