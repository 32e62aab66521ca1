#!/bin/bash
# in-step A/B of the GEMM tile threshold (half-size tiles for the one-round grids: stragglers on the CUs that host an FPS workgroup)
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-probes --steps 60 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['families_ms_per_step'].get('sph3d_pointwise_gemm*'))"; }
for i in 1 2; do
  echo "round $i: default            $(run X=1)"
  echo "round $i: mintiles 1100      $(run SPH3D_SPLIT_MINTILES=1100 SPH3D_GEMM_MINTILES=1100)"
  echo "round $i: mintiles 2100      $(run SPH3D_SPLIT_MINTILES=2100 SPH3D_GEMM_MINTILES=2100)"
  echo "round $i: mintiles 4200      $(run SPH3D_SPLIT_MINTILES=4200 SPH3D_GEMM_MINTILES=4200)"
done
