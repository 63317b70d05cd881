#!/bin/bash
# Round 4, run G: full GPU suite (verbose, per-test timeout, progress in the log), then LBP tile rows x register budget,
# Haar tile rows per step, the fast variance norm factor A/B, host-frame staging threads.
O=gpurun_out/r4_g.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
timeout -k 10 1100 python -m pytest tests/test_gpu_specialize.py tests/test_gpu_split.py tests/test_host_logic.py tests/test_host_sanitizers.py tests/test_oracle_detect.py tests/test_oracle_kats.py -x -v -m gpu --timeout 240 > gpurun_out/r4_t3.log 2>&1
tail -n 4 gpurun_out/r4_t3.log
grep -q " passed" gpurun_out/r4_t3.log && ! grep -q "failed\|Timeout" gpurun_out/r4_t3.log || exit 1
{
echo "### LBP: tile rows x register budget"
for ty in 8 12 16 20 24 32; do for w in 5 6 7; do echo "CCAMD_SPEC_TILE_Y=$ty CCAMD_SPEC_WAVES_PER_EU=$w -- $LBP"; done; done | bash tools/env_sweep.sh
echo "### Haar: fast variance norm factor (default) vs the two rounded operations"
printf 'X=1\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_EXACT_SLOW_VNF\nX=2\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_EXACT_SLOW_VNF -- --content uniform\nX=1 -- --content uniform\n' | bash tools/env_sweep.sh
echo "### Haar: tile rows, STEP-1 tiles only / STEP-2 tiles only"
for ty in 8 12 16 24; do echo "CCAMD_DEBUG_ONLY_STEP=1 CCAMD_SPEC_TILE_Y=$ty"; done | bash tools/env_sweep.sh
for ty in 8 12; do echo "CCAMD_DEBUG_ONLY_STEP=2 CCAMD_SPEC_TILE_Y=$ty"; done | bash tools/env_sweep.sh
echo "### host frames: staging threads"
python tools/time_submit.py 2>/dev/null | tail -n 1
for t in 2 4 8; do CCAMD_STAGE_THREADS=$t python tools/time_submit.py host 2>/dev/null | tail -n 3; done
} > $O 2>&1
tail -n 60 $O
