#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > $O/r05d_tests.log; tail -3 $O/r05d_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2 3; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "round $i: prune $a   no-prune $b" | tee -a $O/r05d_ab.log
done
for c in eval; do
  a=$(timeout 300 python bench.py --eval --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --eval --no-cpu-baseline --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "eval: prune $a   no-prune $b" | tee -a $O/r05d_ab.log
done
for c in modelnet scannet; do
  a=$(timeout 300 python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_FPS_PRUNE=0 timeout 300 python bench.py --config $c --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "$c: prune $a   no-prune $b" | tee -a $O/r05d_ab.log
done
