#!/bin/bash
# usage: gpu_pmc3.sh <tag> <script.py> "<env assignments>" "<counters>" <kernel substring>; prints per-kernel counter averages
TAG=$1; SCRIPT=$2; ENVS=$3; CNT=$4; PAT=$5
export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && env $ENVS timeout -k 5 150 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/$SCRIPT > $GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG.log 2>&1
echo "rc=$?"
python - <<PY
import csv, glob, collections
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "$PAT" in k: acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
