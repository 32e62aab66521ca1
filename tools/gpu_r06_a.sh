#!/bin/bash
# round 6, call A: the ring kernel's tests, the per-shape table (training fused vs separate; inference ring vs barrier), and the
# headline with / without the fused training layers, alternating on one box
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sepring.py -x -q 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_round3.py -x -q -k "fused_inference or separable" 2>&1 | tail -5
( timeout 300 python tools/exp_sepconv_training.py; SPH3D_SC_RING=0 timeout 300 python tools/exp_sepconv_training.py | grep -v "give-ups";
  for nb in 3 4 6; do SPH3D_SR_NB=$nb timeout 300 python tools/exp_sepconv_training.py; done ) > $OUT/r06_exp_sepconv_training.log 2>&1
cat $OUT/r06_exp_sepconv_training.log
for i in 1 2 3; do
  for m in 0 1; do
    v=$(SPH3D_FUSE_TRAIN=$m timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i SPH3D_FUSE_TRAIN=$m: $v" | tee -a $OUT/r06_ab_fuse_train.log
  done
done
