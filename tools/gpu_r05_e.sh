#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $O/r05e_tests.log
for i in 1 2 3; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --no-probes --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  b=$(SPH3D_LIB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_ab.so timeout 300 python bench.py --no-cpu-baseline --no-probes --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
  echo "round $i: zero-fill kernel $a   hipMemsetAsync (previous build) $b" | tee -a $O/r05e_ab.log
done
