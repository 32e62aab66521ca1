#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_fps_prune.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_full_configs.py -m gpu -q -x -k "fps or sampl or full or config" 2>&1 | tail -3
timeout 300 python tools/exp_fps.py 2>&1 | grep -v amdgpu | tee $O/r05f_fps.log
python tools/exp_fps_prof.py 2>&1 | grep -v amdgpu | head -20 | tee $O/r05f_fps_prof.log
