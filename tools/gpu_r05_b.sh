#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fps_prune.py tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_full_configs.py -m gpu -q -x -k "fps or loss_kernel or sampl or full or config" 2>&1 | tail -15 > $O/r05b_tests.log; tail -3 $O/r05b_tests.log
timeout 300 python tools/exp_fps.py > $O/r05b_fps.log 2>&1; cat $O/r05b_fps.log | tail -14
for i in 1 2; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  c=$(SPH3D_FPS_NW4_MAX=8192 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "round $i: prune $a   no-prune $b   prune-nw4-8192 $c" | tee -a $O/r05b_ab.log
done
a=$(timeout 300 python bench.py --eval --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --eval --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
echo "eval: prune $a   no-prune $b" | tee -a $O/r05b_ab.log
