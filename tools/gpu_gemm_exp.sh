#!/bin/bash
# GEMM phase-skip diagnostics: the product kernels with parts of the k loop compiled out (-DSPH3D_GEMM_EXP: 1 no global loads,
# 2 no LDS stores, 4 no barrier; results are wrong, only the time means something).  Builds the variants next to the in-tree
# library on the GPU box.  -DSPH3D_GEMM_DMA=0 selects the register-staged loop the switches 1 and 2 were written for.
cd $GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function"
OBJS="graph.o tile.o convtile.o nnquery.o buildkernel.o conv3d.o pool3d.o sample.o norm.o sepconv.o api.o"
for e in 0 1 3 7; do
  /opt/rocm/bin/hipcc $FLAGS -DSPH3D_GEMM_DMA=0 -DSPH3D_GEMM_EXP=$e -c gemm.hip -o /tmp/gemm_e$e.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_e$e.o -o /tmp/libsph3d_exp$e.so
done
cd $GRAFT_REPO_ROOT
echo "=== in-tree (LDS-DMA)"; python tools/exp_gemm.py 2>&1 | grep -E "R131072 Cin  (128|256) Cout 128|R 32768 Cin  512|total" | cut -c1-95
for e in 0 1 3 7; do
  echo "=== register-staged loop, GEMM_EXP=$e"
  SPH3D_LIB=/tmp/libsph3d_exp$e.so python tools/exp_gemm.py 2>&1 | grep -E "R131072 Cin  (128|256) Cout 128|R 32768 Cin  512|total" | cut -c1-95
done
