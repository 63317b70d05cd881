#!/bin/bash
# Round 4, run B: counters of the LBP kernel with the bank-class table everywhere vs the list from stage 2 on; lean split search.
O=gpurun_out/r4_b.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > gpurun_out/r4_counters_list.txt 2>&1
echo "### split search: tests, then lean vs branchy"
timeout -k 10 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_config_sizes.py -x -q -m gpu 2>&1 | tail -n 3
python tools/bench_split_search.py HAAR 20000
CCAMD_SPLIT_BRANCHY=1 python tools/bench_split_search.py HAAR 20000
echo "### LBP counters: class table everywhere (CCAMD_DENSE_FROM=0) vs list from stage 2 (default)"
for v in 0 2; do
export CCAMD_DENSE_FROM=$v
echo "--- CCAMD_DENSE_FROM=$v"
bash tools/pmc_any.sh lbpd1 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" $LBP
bash tools/pmc_any.sh lbpd2 "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" $LBP
done
unset CCAMD_DENSE_FROM
echo "### stamps, list from stage 2"
echo "X=1 -- $LBP" | bash tools/stamp_compare.sh
} > $O 2>&1
tail -n 40 $O
