#!/bin/bash
# round 6, call C: GEMM with three tile images in flight (SPH3D_GEMM_NBUF=3) against the double buffer (libsph3d_nbuf2.so), per
# shape and in the step; and the transposed-graph merge against the commit before it (build_exp/base)
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "gemm or Gemm or net or step" 2>&1 | tail -3
( echo "== NBUF 3 (in-tree)"; timeout 300 python tools/exp_gemm.py; echo "== NBUF 2"; SPH3D_LIB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_nbuf2.so timeout 300 python tools/exp_gemm.py ) 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_exp_gemm_nbuf.log
for i in 1 2 3; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  b=$(SPH3D_LIB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_nbuf2.so timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  c=$(cd build_exp/base && timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/tmp/ab_base.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $i: tree (tg merge + NBUF 3) $a | tree with NBUF 2 $b | base (r05 end) $c" | tee -a $OUT/r06_ab_tg_nbuf.log
done
tail -3 /tmp/ab_base.err
