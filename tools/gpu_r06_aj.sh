#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2 3; do for f in 1 0; do
    v=$(SPH3D_FUSE_POOL_GRAPH=$f timeout 300 python bench.py --no-cpu-baseline --steps 80 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "headline round $i: SPH3D_FUSE_POOL_GRAPH=$f: $v"
done; done | tee $OUT/r06_ab_pool_graph.log
