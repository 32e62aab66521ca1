#!/bin/bash
# SQ counter passes of the LDS-tile forward kernel at S3DIS level 0: usage gpu_pmc_lds.sh "<extra env>" <tag>
X="$1"; T=${2:-lds}
S1="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES"
S2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
S3="GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"
bash tools/gpu_pmc3.sh ${T}_a tools/exp_conv_pmc.py "LDS=1 $X" "$S1" dwconv_fwd_lds
bash tools/gpu_pmc3.sh ${T}_b tools/exp_conv_pmc.py "LDS=1 $X" "$S2" dwconv_fwd_lds
bash tools/gpu_pmc3.sh ${T}_c tools/exp_conv_pmc.py "LDS=1 $X" "$S3" dwconv_fwd_lds
