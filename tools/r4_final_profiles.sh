#!/bin/bash
# Round 4 evidence for the committed kernels (run on the GPU box; results land in gpurun_out/r4_profiles/, copy them to profiles/):
#   r04_final_kernel_stats.csv / r04_final_summary.json / r04_traffic_k_eval.json / r04_pmc_eval.json      (headline Haar)
#   r04_lbp_final_kernel_stats.csv / r04_lbp_final_summary.json / r04_traffic_k_eval_lbp.json / r04_pmc_eval_lbp.json (configs[2])
#   r04_block_stamps.txt (per-phase block stamps of both kernels), r04_training_kernels.txt (k_split_ord counters)
set -o pipefail
D=gpurun_out/r4_profiles
mkdir -p $D profiles
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
echo "[1/6] headline stats + traffic"; bash tools/profile_bench.sh r04 --steps 20 --warmup 5 > $D/log_prof_haar.txt 2>&1
python tools/summarize_profile.py r04 r04_final > $D/log_sum_haar.txt 2>&1
echo "[2/6] headline counters"; bash tools/pmc_eval.sh r04 > $D/log_pmc_haar.txt 2>&1; cp gpurun_out/pmc_r04/summary.json profiles/r04_pmc_eval.json
echo "[3/6] LBP stats + traffic"; bash tools/profile_bench.sh r04lbp --steps 20 --warmup 5 $LBP > $D/log_prof_lbp.txt 2>&1
python tools/summarize_profile.py r04lbp r04_lbp_final _lbp > $D/log_sum_lbp.txt 2>&1
echo "[4/6] LBP counters"; bash tools/pmc_eval.sh r04lbp $LBP > $D/log_pmc_lbp.txt 2>&1; cp gpurun_out/pmc_r04lbp/summary.json profiles/r04_pmc_eval_lbp.json
echo "[5/6] block stamps"
{ echo "== LBP (stock cascade, 20 stages compiled, tiles of 20 window rows, list queue from stage 2)"; echo "X=1 -- $LBP" | bash tools/stamp_compare.sh
  echo "== Haar (headline cascade, 7 stages compiled, one module per step)"; echo "X=1" | bash tools/stamp_compare.sh; } > $D/r04_block_stamps.txt 2>&1
echo "[6/6] split search counters"
{ echo "k_split_ord / k_split_ord_lean at configs[4] (tools/bench_split_search.py HAAR 20000), rocprofv3 --pmc, averages per launch (M = 1e6)"
  bash tools/pmc_cmd.sh split1 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" k_split tools/bench_split_search.py HAAR 20000
  bash tools/pmc_cmd.sh split2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAVES" k_split tools/bench_split_search.py HAAR 20000
  echo "-- the round-1 kernel (CCAMD_SPLIT_BRANCHY=1)"
  CCAMD_SPLIT_BRANCHY=1 bash tools/pmc_cmd.sh split3 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" k_split tools/bench_split_search.py HAAR 20000
  python tools/bench_split_search.py HAAR 20000 2>/dev/null; CCAMD_SPLIT_BRANCHY=1 python tools/bench_split_search.py HAAR 20000 2>/dev/null; } > $D/r04_training_kernels.txt 2>&1
cp profiles/r04_* $D/ 2>/dev/null
ls -la $D
