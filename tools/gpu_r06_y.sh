#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for cfg in "16384 1024" "32768 1024" "100000000 1024"; do
    set -- $cfg
    v=$(SPH3D_BWD_HUB_MIN_N=$1 SPH3D_BWD_HUB_T=$2 timeout 400 python bench.py --config scannet --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['families_ms_per_step'].get('sph3d_depthwise_conv3d_grad_t'), d['families_ms_per_step'].get('sph3d_depthwise_conv3d_grad_t_cat'))")
    echo "scannet round $i: hub min N $1, threshold $2: $v"
  done
done | tee $OUT/r06_ab_conv_hub.log
