#!/bin/bash
# round 6, call G: split-bf16 GEMM — tests, per-shape table against the fp32-MFMA kernels and rocBLAS, headline A/B
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -15
( echo "== split bf16x6 (default)"; timeout 300 python tools/exp_gemm.py; echo "== fp32 MFMA (SPH3D_GEMM_SPLIT=0)"; SPH3D_GEMM_SPLIT=0 timeout 300 python tools/exp_gemm.py ) 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_exp_gemm_split.log
for i in 1 2 3; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  b=$(SPH3D_GEMM_SPLIT=0 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $i: split $a | fp32 MFMA $b" | tee -a $OUT/r06_ab_gemm_split.log
done
