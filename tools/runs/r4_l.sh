#!/bin/bash
# Round 4, run L: would TALL 16-bit STEP-2 tiles pay for Haar? Existing CCAMD_SPEC_TILE16 build (table-driven stages and the wave
# phase read global memory), STEP-2 tiles only, stopped after the compiled stages (stage 6) -- against the shipped 32-bit tiles.
O=gpurun_out/r4_l.txt
{
echo "### STEP-2 tiles only, through stage 6: shipped 32-bit tiles (8 rows, module per step)"
printf 'CCAMD_DEBUG_ONLY_STEP=2 CCAMD_DEBUG_STOP_AFTER_STAGE=6\nCCAMD_DEBUG_ONLY_STEP=2\n' | bash tools/env_sweep.sh
echo "### the same with 16-bit tiles, rows x wavefronts per SIMD"
for ty in 8 12 16 20; do for w in 5 6; do echo "CCAMD_SPEC_TILE16=1 CCAMD_DEBUG_ONLY_STEP=2 CCAMD_DEBUG_STOP_AFTER_STAGE=6 CCAMD_SPEC_TILE_Y=$ty CCAMD_SPEC_WAVES_PER_EU=$w"; done; done | bash tools/env_sweep.sh
echo "### 16-bit tiles, whole cascade (wave phase on global memory)"
for ty in 8 16; do echo "CCAMD_SPEC_TILE16=1 CCAMD_DEBUG_ONLY_STEP=2 CCAMD_SPEC_TILE_Y=$ty"; done | bash tools/env_sweep.sh
} > $O 2>&1
cat $O
