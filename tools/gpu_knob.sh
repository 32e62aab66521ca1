#!/bin/bash
# bench under a list of environment settings, alternating, two rounds: step time + GEMM / conv-gradient family times
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for v in "$@"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); f=d['families_ms_per_step']; print('%-34s %.3f ms  gemm %.2f  convgrad %.2f  conv %.2f graph %.2f' % ('$v', d['ms_per_step'], f['sph3d_pointwise_gemm*'], f['sph3d_depthwise_conv3d_grad_t'], f['sph3d_depthwise_conv3d'], f['sph3d_build_sphere_graph']))"
done; done
