#!/bin/bash
for m in plain; do
bash tools/gpu_pmc3.sh nn1_$m tools/exp_nn_pmc.py "NN=$m" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" nnquery | tail -2
bash tools/gpu_pmc3.sh nn2_$m tools/exp_nn_pmc.py "NN=$m" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" nnquery | tail -2
bash tools/gpu_pmc3.sh nn3_$m tools/exp_nn_pmc.py "NN=$m" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" nnquery | tail -2
done
