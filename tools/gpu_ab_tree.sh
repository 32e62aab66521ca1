#!/bin/bash
# A/B of two source trees inside ONE gpurun call (same box, alternating runs): the working tree against build_exp/base (a
# `git archive` of the commit to compare with, its library built in place).  usage: bash tools/gpu_ab_tree.sh [rounds] [steps] [bench args]
R=${1:-3}; S=${2:-80}; shift; shift
export TMPDIR=/tmp
for i in $(seq $R); do
  a=$(cd $GRAFT_REPO_ROOT && timeout 300 python bench.py --no-cpu-baseline --steps $S "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  b=$(cd $GRAFT_REPO_ROOT/build_exp/base && timeout 300 python bench.py --no-cpu-baseline --steps $S "$@" 2>/tmp/ab_base.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "round $i: working tree $a   base $b"
  [ -z "$b" ] && tail -5 /tmp/ab_base.err
done
