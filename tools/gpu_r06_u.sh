#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
AB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_ab.so
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -4
for i in 1 2 3; do
  a=$(timeout 300 python bench.py --config modelnet --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  b=$(SPH3D_LIB=$AB timeout 300 python bench.py --config modelnet --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "modelnet round $i: ragged shapes on the bf16 pipe $a | fp32 kernels for ragged shapes $b"
done | tee $OUT/r06_ab_gemm_split_guard.log
a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
b=$(SPH3D_LIB=$AB timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
echo "headline: $a | $b" | tee -a $OUT/r06_ab_gemm_split_guard.log
