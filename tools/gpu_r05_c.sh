#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
for wh in 1 0; do SPH3D_FPS_WHOLE=$wh timeout 600 python -m pytest tests/test_gpu_fps_prune.py tests/test_gpu_parity.py -m gpu -q -x -k "fps" 2>&1 | tail -3; done > $O/r05c_tests.log; cat $O/r05c_tests.log
timeout 300 python tools/exp_fps.py > $O/r05c_fps.log 2>&1; grep -v amdgpu $O/r05c_fps.log
SPH3D_FPS_WHOLE=0 python tools/exp_fps_prof.py 2>&1 | grep -v amdgpu | head -20 | tee $O/r05c_fps_prof.log
