#!/bin/bash
# Round 4, run S: tile staging dealt to threads as one list of (row, group) pairs (default) / a wavefront per row (-DCC_STAGE_BY_ROWS).
O=gpurun_out/r4_s.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
echo "### Haar"
printf 'X=1\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_STAGE_BY_ROWS\nX=2\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_STAGE_BY_ROWS\n' | bash tools/env_sweep.sh
echo "### LBP"
printf "X=1 -- $LBP\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_STAGE_BY_ROWS -- $LBP\nX=2 -- $LBP\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_STAGE_BY_ROWS -- $LBP\n" | bash tools/env_sweep.sh
echo "### uniform frames, Haar"
printf 'X=1 -- --content uniform\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_STAGE_BY_ROWS -- --content uniform\n' | bash tools/env_sweep.sh
echo "### LBP split search tail (CCAMD_TRACE_SPLIT)"
CCAMD_TRACE_SPLIT=1 python tools/bench_split_search.py LBP 20000 2>&1 | grep "ccamd split" | tail -4
} > $O 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_specialize.py tests/test_gpu_detect.py tests/test_gpu_detect_variants.py tests/test_gpu_random_parity.py -q -m gpu --timeout 300 > gpurun_out/r4_s_tests.log 2>&1
tail -n 3 gpurun_out/r4_s_tests.log >> $O
cat $O
