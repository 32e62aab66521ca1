#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -2
timeout 300 python tools/exp_gemm.py 2>&1 | grep -v amdgpu.ids | sed "s/| blas.*//" | tee $OUT/r06_exp_gemm_split_swz.log
for kind in nn tn; do bash tools/gpu_pmc3.sh sw_$kind tools/exp_gemm_pmc.py "KIND=$kind SHAPE=6144,2048,256" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" gemm_split | tail -1; done | tee -a $OUT/r06_exp_gemm_split_swz.log
bash tools/gpu_ab_tree.sh 2 80 | tee -a $OUT/r06_exp_gemm_split_swz.log
