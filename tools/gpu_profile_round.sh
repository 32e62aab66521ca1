#!/bin/bash
# One gpurun call that produces the artefacts of a round under gpurun_out/<tag>_*: default bench line, rocprofv3 kernel
# stats + per-launch-shape durations of the same command, and separate PMC passes (traffic, then L2 hit) of the conv kernels.
# usage: bash tools/gpu_profile_round.sh r02
TAG=${1:?usage: bash tools/gpu_profile_round.sh <tag, e.g. r06>}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $OUT/${TAG}_bench_default.log 2> $OUT/${TAG}_bench_default.err; echo "bench rc=$?"
# secondary lines (BASELINE configs 2, 3, 5 and the inference forward): same contract, not the headline
for c in modelnet shapenet scannet; do timeout 400 python bench.py --config $c --steps 20 --warmup 3 > $OUT/${TAG}_bench_$c.log 2> $OUT/${TAG}_bench_$c.err; done
timeout 400 python bench.py --eval --steps 20 --warmup 3 > $OUT/${TAG}_bench_eval.log 2> $OUT/${TAG}_bench_eval.err
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
        key = (r["Kernel_Name"][:90], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = sorted(acc.items(), key=lambda kv: -sum(kv[1]))
    with open(out + "/%s_kernel_trace_by_launch_shape.csv" % tag, "w") as f:
        f.write("kernel,grid_x,wg_x,calls,total_us,mean_us,min_us,max_us\n")
        for (k, g, w), v in rows[:120]:
            f.write('"%s",%s,%s,%d,%.1f,%.1f,%.1f,%.1f\n' % (k, g, w, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
PY
# PROFILE_LIGHT=1: skip the conv / GEMM counter passes (their kernels did not change since the last full run)
if [ -z "$PROFILE_LIGHT" ]; then
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $pass | cut -c1-3)
  for cfg in "fwd:" "bwd:"; do
    name=${cfg%%:*}; env=${cfg#*:}; mode=fwd; [ "$name" = "bwd" ] && mode=bwd
    bash tools/gpu_pmc3.sh ${TAG}_${name}_$n tools/exp_conv_pmc.py "$env MODE=$mode C=128" "$pass" dwconv | tail -2
  done
done
# SQ counters: the conv gather / gradient (instruction mix, waits) and the GEMMs (MFMA pipe busy cycles)
for cfg in "fwd:MODE=fwd" "bwd:MODE=bwd"; do
  name=${cfg%%:*}; env=${cfg#*:}
  bash tools/gpu_pmc3.sh ${TAG}_sq1_$name tools/exp_conv_pmc.py "$env C=128" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" dwconv | tail -2
  bash tools/gpu_pmc3.sh ${TAG}_sq2_$name tools/exp_conv_pmc.py "$env C=128" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" dwconv | tail -2
done
for cfg in "nn:KIND=nn SHAPE=131072,256,128" "nt:KIND=nt SHAPE=131072,256,128" "tn:KIND=tn SHAPE=32768,1024,128" "tn0:KIND=tn SHAPE=131072,256,128"; do
  name=${cfg%%:*}; env=${cfg#*:}
  bash tools/gpu_pmc3.sh ${TAG}_mfma_$name tools/exp_gemm_pmc.py "$env" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE" gemm | tail -3
done
ls $OUT | grep "^$TAG" | head -60
# HBM traffic of the GEMMs the roofline object may name (separate passes, as for the conv kernels)
for cfg in "nn:KIND=nn SHAPE=131072,256,128" "tn:KIND=tn SHAPE=32768,1024,128" "tn0:KIND=tn SHAPE=131072,256,128"; do
  name=${cfg%%:*}; env=${cfg#*:}
  for pass in "FETCH_SIZE" "WRITE_SIZE"; do
    n=$(echo $pass | cut -c1-3)
    bash tools/gpu_pmc3.sh ${TAG}_gemm${name}_$n tools/exp_gemm_pmc.py "$env" "$pass" gemm | tail -3
  done
done
fi
# neighbour search (level-0 plain search, 16 x 8192, K = 64): instruction mix of the in-tree kernels (grid build / order / search,
# dense scan; SPH3D_NNGRID=0: the chain kernel alone) and, when a library built
# with the previous scan is present (sph3d_gcn_amd/csrc/libsph3d_nnbefore.so), of that one.  To build it: `git show 382d868~1:sph3d_gcn_amd/csrc/nnquery.hip`
# into a scratch directory next to copies of common.hpp / sphere_bin.hpp, hipcc -c it with the Makefile's flags and link it with the
# in-tree objects of the other sources (the library is git-ignored and was not kept)
for cfg in "after:" "chain:SPH3D_NNGRID=0" "before:SPH3D_LIB=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_nnbefore.so"; do
  name=${cfg%%:*}; env=${cfg#*:}
  [ "$name" = "before" ] && [ ! -f $GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_nnbefore.so ] && continue
  bash tools/gpu_pmc3.sh ${TAG}_nnquery_$name tools/exp_nn_pmc.py "NN=plain $env" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH GRBM_GUI_ACTIVE" nn | tail -6
done
# one flat CSV per counter pass next to the summaries (what gets copied into profiles/)
for d in $OUT/pmc_${TAG}_*; do
  [ -d "$d" ] || continue
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/$(basename $d | sed "s/^pmc_${TAG}_/${TAG}_pmc_/").csv
done
ls $OUT | grep "^$TAG" | head -80
