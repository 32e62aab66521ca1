#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_ab.so
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -3
( echo "== in-tree"; timeout 300 python tools/exp_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/| blas.*//'
  echo "== ab copy"; SPH3D_LIB=$AB timeout 300 python tools/exp_gemm.py 2>&1 | grep -v amdgpu.ids | sed 's/| blas.*//' ) | tee $OUT/r06_exp_gemm_split_ab.log
for i in 1 2 3; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  b=$(SPH3D_LIB=$AB timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  c=$(SPH3D_GEMM_SPLIT=0 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $i: in-tree $a | ab copy $b | fp32 MFMA $c"
done | tee -a $OUT/r06_exp_gemm_split_ab.log
