#!/bin/bash
# A/B of an environment switch inside ONE gpurun call (same box, alternating runs).  usage: bash tools/gpu_ab_env.sh VAR=a VAR=b [rounds] [steps]
A=$1; B=$2; R=${3:-3}; S=${4:-80}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  a=$(env $A timeout 300 python bench.py --no-cpu-baseline --steps $S 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(env $B timeout 300 python bench.py --no-cpu-baseline --steps $S 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "round $i: $A $a   $B $b"
done
