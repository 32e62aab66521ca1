#!/bin/bash
# round 5, call A: GPU tests, FPS pruning times, gradient source orders, the scatter-reduce prices, headline A/B of the pruning
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/r05a_tests.log; echo "tests rc=$?"; tail -3 $O/r05a_tests.log
timeout 300 python tools/exp_fps.py > $O/r05a_fps.log 2>&1; cat $O/r05a_fps.log | tail -14
timeout 120 tools/micro/scatter_reduce > $O/r05a_scatter.log 2>&1; cat $O/r05a_scatter.log
timeout 600 python tools/exp_bwd_order.py > $O/r05a_bwd_order.log 2>&1; tail -45 $O/r05a_bwd_order.log
for i in 1 2; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "round $i: prune $a   no-prune $b" | tee -a $O/r05a_ab.log
done
a=$(timeout 300 python bench.py --eval --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --eval --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
echo "eval: prune $a   no-prune $b" | tee -a $O/r05a_ab.log
