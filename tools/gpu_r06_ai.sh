#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for pk in 1 0; do
    v=$(SPH3D_TG_PACK=$pk timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['families_ms_per_step'].get('sph3d_graph_transpose_finish_ordered'))")
    echo "headline round $i: SPH3D_TG_PACK=$pk: $v"
  done
done | tee $OUT/r06_ab_tg_pack.log
for c in modelnet shapenet; do for i in 1 2; do for pk in 1 0; do
    v=$(SPH3D_TG_PACK=$pk timeout 300 python bench.py --config $c --steps 20 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['families_ms_per_step'].get('sph3d_graph_transpose_finish_ordered'))")
    echo "$c round $i: SPH3D_TG_PACK=$pk: $v"
done; done; done | tee -a $OUT/r06_ab_tg_pack.log
