#!/bin/bash
# usage: gpu_pmc2.sh <tag> "<env assignments>" "<counters>" ; prints the per-kernel counter averages of the conv kernels
TAG=$1; ENVS=$2; CNT=$3
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && env $ENVS timeout -k 5 150 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/tools/exp_conv_pmc.py > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1
echo "rc=$?"
python - <<PY
import csv, glob, collections
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "dwconv" in k: acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
