#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc
( for P in 4 8 16; do lib=$L/libsph3d_p$P.so; [ $P = 4 ] && lib=$L/libsph3d.so
  TAG="P$P" SPH3D_LIB=$lib timeout 300 python tools/exp_fps_coop.py 2>&1 | grep -v amdgpu.ids | head -3
  SPH3D_LIB=$lib timeout 300 python -m pytest tests -m gpu -x -q -k "fps or FPS or sample" 2>&1 | tail -1
done
for i in 1 2; do for P in 4 8 16; do lib=$L/libsph3d_p$P.so; [ $P = 4 ] && lib=$L/libsph3d.so
  v=$(SPH3D_LIB=$lib timeout 400 python bench.py --config scannet --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['families_ms_per_step'].get('sph3d_farthest_point_sample'))")
  echo "scannet round $i: co-operative FPS with $P points per thread: $v"
done; done ) | tee $OUT/r06_ab_fps_coop_p.log
