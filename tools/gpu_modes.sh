#!/bin/bash
mkdir -p gpurun_out
for args in "" "--gemm blas"; do
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline $args 2>gpurun_out/err.log | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$args', '|', d['value'],'blocks/s', d['ms_per_step'],'ms', d['config']['launch_mode'], d['config']['gemm_backend'], 'sph3d_ms', d['sph3d_kernels_ms_per_step'])
"
  grep -i "capture failed" gpurun_out/err.log | head -2
done
