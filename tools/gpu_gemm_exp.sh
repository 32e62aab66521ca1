#!/bin/bash
# GEMM phase-skip diagnostics: the product kernels with parts of the k loop compiled out (results are wrong; time only)
cd $GRAFT_REPO_ROOT
for e in "" 1 3 7; do
  lib=""; [ -n "$e" ] && lib=$GRAFT_REPO_ROOT/sph3d_gcn_amd/csrc/libsph3d_exp$e.so
  echo "=== GEMM_EXP=${e:-0}"
  SPH3D_LIB=$lib python tools/exp_gemm.py 2>&1 | grep -E "R131072 Cin  (128|256) Cout 128|R 32768 Cin  512|total" | cut -c1-95
done
