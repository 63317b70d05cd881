#!/bin/bash
# Round-4 opening measurements: counters + block stamps + stop-after profile of the shipped LBP cascade kernel
# (configs[2]) and counters of k_split_ord. Output: gpurun_out/r4_base.txt
set -o pipefail
O=gpurun_out/r4_base.txt
mkdir -p gpurun_out
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
echo "### LBP k_eval_spec counters (tools/pmc_any.sh: --device-only --frames 32, no front-end overlap)"
bash tools/pmc_any.sh lbp1 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" $LBP
bash tools/pmc_any.sh lbp2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_IFETCH SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" $LBP
bash tools/pmc_any.sh lbp3 "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES" $LBP
bash tools/pmc_any.sh lbp_rd "FETCH_SIZE" $LBP
bash tools/pmc_any.sh lbp_wr "WRITE_SIZE" $LBP
echo "### LBP block stamps"
echo "X=1 -- $LBP" | bash tools/stamp_compare.sh
echo "### LBP stop-after profile (ms per 32 frames alone)"
for s in -2 0 1 2 3 4 5 6 7 8 10 19; do echo "CCAMD_DEBUG_STOP_AFTER_STAGE=$s -- $LBP"; done | bash tools/env_sweep.sh
echo "### Haar headline kernel alone"
echo "X=1" | bash tools/env_sweep.sh
echo "### k_split_ord counters (tools/bench_split_search.py HAAR 20000)"
bash tools/pmc_cmd.sh split1 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" k_split tools/bench_split_search.py HAAR 20000
bash tools/pmc_cmd.sh split2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES" k_split tools/bench_split_search.py HAAR 20000
bash tools/pmc_cmd.sh split_rd "FETCH_SIZE" k_split tools/bench_split_search.py HAAR 20000
} > $O 2>&1
tail -n 60 $O
