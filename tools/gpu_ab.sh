#!/bin/bash
# A/B of two builds of libsph3d.so inside ONE gpurun call (same box, alternating runs): the in-tree library against
# sph3d_gcn_amd/csrc/libsph3d_ab.so (a copy of the build to compare with).  usage: bash tools/gpu_ab.sh [rounds] [steps]
R=${1:-3}; S=${2:-80}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  a=$(timeout 300 python bench.py --no-cpu-baseline --steps $S 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_LIB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_ab.so timeout 300 python bench.py --no-cpu-baseline --steps $S 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "round $i: in-tree $a   ab-copy $b"
done
