#!/bin/bash
# one gpurun call: parity tests of the LDS-tile forward, then isolated timings with the phase switches
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_convlds.py -x -q -m gpu 2>&1 | tail -15

for w in 16; do for d in 0 1 2 4; do SPH3D_LC_WAVES=$w SPH3D_LC_DBG=$d timeout 300 python tools/exp_convlds.py $([ $d != 0 ] && echo l0); done; done
} > gpurun_out/convlds.log 2>&1
tail -70 gpurun_out/convlds.log
