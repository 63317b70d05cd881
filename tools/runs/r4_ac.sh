#!/bin/bash
# Round 4, run AC: LBP wave phase, stage sums from eight votes per LDS round trip (default) / one per trip (-DCC_LBP_WAVE_SERIAL_SUM); parity.
O=gpurun_out/r4_ac.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
printf "X=1 -- $LBP\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_LBP_WAVE_SERIAL_SUM -- $LBP\nX=2 -- $LBP\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_LBP_WAVE_SERIAL_SUM -- $LBP\nX=3 -- $LBP --content uniform\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_LBP_WAVE_SERIAL_SUM -- $LBP --content uniform\n" | bash tools/env_sweep.sh
} > $O 2>&1
timeout -k 10 800 python -m pytest tests/test_gpu_specialize.py tests/test_gpu_detect.py tests/test_gpu_config_sizes.py -q -m gpu --timeout 300 -k "lbp or LBP or queue or tile" > gpurun_out/r4_ac_tests.log 2>&1
tail -n 3 gpurun_out/r4_ac_tests.log >> $O
cat $O
