#!/bin/bash
# Round 4, run Z: what a 16-bit integral could buy the LBP kernel's staging (timing only: -DCC_DEBUG_STAGE_HALF reads half the bytes).
# Staging alone (stop after -2) and staging + dense phase (-3 is Haar only; stage 0 = 0), where the wrong tile contents cannot change the work.
O=gpurun_out/r4_z.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
for s in -2 0; do
printf "CCAMD_DEBUG_STOP_AFTER_STAGE=$s -- $LBP\nCCAMD_DEBUG_STOP_AFTER_STAGE=$s CCAMD_SPEC_EXTRA_FLAGS=-DCC_DEBUG_STAGE_HALF -- $LBP\n" | bash tools/env_sweep.sh
done
echo "### Haar, staging only"
printf "CCAMD_DEBUG_STOP_AFTER_STAGE=-2\nCCAMD_DEBUG_STOP_AFTER_STAGE=-2 CCAMD_SPEC_EXTRA_FLAGS=-DCC_DEBUG_STAGE_HALF\n" | bash tools/env_sweep.sh
} > $O 2>&1
cat $O
