#!/bin/bash
# exchange-split A/B: tests, per-shape table with and without, headline A/B via env
cd $GRAFT_REPO_ROOT
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_gemm_xs.py tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -5
{
echo "== exchange split on (default)"; timeout 300 python tools/exp_gemm.py
echo "== SPH3D_GEMM_XS=0 SPH3D_GEMM_TN_XS=0"; SPH3D_GEMM_XS=0 SPH3D_GEMM_TN_XS=0 timeout 300 python tools/exp_gemm.py
echo "== SPH3D_GEMM_XS=4 SPH3D_GEMM_TN_XS=6144"; SPH3D_GEMM_XS=4 SPH3D_GEMM_TN_XS=6144 timeout 300 python tools/exp_gemm.py
} > $OUT/r06_exp_gemm_xs.log 2>&1
grep -v "^R131072\|^R 32768" $OUT/r06_exp_gemm_xs.log
for i in 1 2; do
  a=$(timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probes | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  b=$(SPH3D_GEMM_XS=0 SPH3D_GEMM_TN_XS=0 timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probes | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  echo "round $i: xs $a   off $b" | tee -a $OUT/r06_exp_gemm_xs.log
done
