#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_fps_prune.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_full_configs.py -m gpu -q -x -k "fps or sampl or full or config" 2>&1 | tail -3
echo "== runner-up"; SPH3D_FPS_PRUNE=2049 python tools/exp_fps.py child 2>&1 | grep -v amdgpu | tee $O/r05f_fps_runnerup.log
echo "== full elections only"; SPH3D_LIB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_ab.so SPH3D_FPS_PRUNE=2049 python tools/exp_fps.py child 2>&1 | grep -v amdgpu | tee -a $O/r05f_fps_runnerup.log
