#!/bin/bash
# first GPU round: parity tests, smoke, golden vectors from the reference build, short bench, rocprof stats
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.cuda.get_device_name(0))" > gpurun_out/device.txt 2>&1
nproc >> gpurun_out/device.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> gpurun_out/device.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 300 python tests/golden/make_golden.py gpurun_out/golden > gpurun_out/golden.log 2>&1; echo "golden rc=$?" >> gpurun_out/golden.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/prof.log
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
