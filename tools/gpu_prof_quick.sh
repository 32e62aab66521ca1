#!/bin/bash
# rocprofv3 kernel stats of 10 headline steps -> gpurun_out/<tag>_kernel_stats.csv (+ per-launch-shape trace)
TAG=${1:?tag}; OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-probes > $OUT/${TAG}_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, shutil
out = "$OUT"; tag = "$TAG"
st = glob.glob(out + "/prof_%s/**/*kernel_stats.csv" % tag, recursive=True)
if st: shutil.copy(st[0], out + "/%s_kernel_stats.csv" % tag)
tr = glob.glob(out + "/prof_%s/**/*kernel_trace.csv" % tag, recursive=True)
if tr:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(tr[0])):
        key = (r["Kernel_Name"][:110], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open(out + "/%s_kernel_trace_by_launch_shape.csv" % tag, "w") as f:
        f.write("kernel,grid_x,wg_x,calls,total_us,mean_us,min_us,max_us\n")
        for (k, g, w), v in rows[:160]:
            f.write('"%s",%s,%s,%d,%.1f,%.1f,%.1f,%.1f\n' % (k, g, w, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
PY
rm -rf $OUT/prof_$TAG
