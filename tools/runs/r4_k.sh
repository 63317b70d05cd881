#!/bin/bash
# Round 4, run K: rolled dense loop of the 20-row LBP tiles (spills 25 -> 9 VGPRs), wave-phase threshold and stage groups at 20 rows.
O=gpurun_out/r4_k.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
timeout -k 10 900 python -m pytest tests/test_gpu_specialize.py tests/test_gpu_config_sizes.py tests/test_gpu_detect.py tests/test_bench_cli.py -x -v -m gpu --timeout 300 > gpurun_out/r4_t6.log 2>&1
tail -n 3 gpurun_out/r4_t6.log
grep -q " passed" gpurun_out/r4_t6.log && ! grep -q "failed\|Timeout" gpurun_out/r4_t6.log || exit 1
{
echo "### LBP at 20 rows: wave-phase threshold, groups, register budget"
printf 'X=1 -- %s\nCCAMD_WAVE_BELOW=16 -- %s\nCCAMD_WAVE_BELOW=32 -- %s\nCCAMD_WAVE_BELOW=48 -- %s\nCCAMD_WAVE_BELOW=64 -- %s\nCCAMD_GROUP_STUMPS=14 -- %s\nCCAMD_GROUP_STUMPS=30 -- %s\nCCAMD_WAVE_BELOW=48 CCAMD_GROUP_STUMPS=14 -- %s\nCCAMD_SPEC_WAVES_PER_EU=7 -- %s\nCCAMD_SPEC_WAVES_PER_EU=5 -- %s\nCCAMD_SPEC_TILE_Y=24 CCAMD_SPEC_WAVES_PER_EU=5 -- %s\nCCAMD_DENSE_FROM=1 -- %s\nCCAMD_DENSE_FROM=3 -- %s\n' "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" | bash tools/env_sweep.sh
echo "### Haar default"
echo "X=1" | bash tools/env_sweep.sh
} > $O 2>&1
tail -n 30 $O
