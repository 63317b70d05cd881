#!/bin/bash
# Round 4, run F: full GPU suite with the 16-row LBP tiles and the fast variance norm factor, then their sweeps.
O=gpurun_out/r4_f.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
echo "### full GPU suite"
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -n 6
echo "### LBP: tile rows x register budget"
for ty in 8 12 16 20 24 32; do for w in 5 6 7; do echo "CCAMD_SPEC_TILE_Y=$ty CCAMD_SPEC_WAVES_PER_EU=$w -- $LBP"; done; done | CCAMD_TRACE_HOST=1 bash tools/env_sweep.sh 2>&1 | grep -v "^\[ccamd host\] pass"
echo "### Haar: fast variance norm factor (default) vs the two rounded operations"
printf 'X=1\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_EXACT_SLOW_VNF\nX=2\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_EXACT_SLOW_VNF -- --content uniform\nX=1 -- --content uniform\n' | bash tools/env_sweep.sh
} > $O 2>&1
tail -n 45 $O
