#!/bin/bash
# Round 4, run R: record prefetch in the Haar wave phase (A/B through CCAMD_SPEC_EXTRA_FLAGS), parity tests of the specialised kernels.
O=gpurun_out/r4_r.txt
{
echo "### Haar wave phase: records prefetched (default) / waited for (-DCC_WAVE_NO_PREFETCH), twice each"
printf 'X=1\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_WAVE_NO_PREFETCH\nX=2\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_WAVE_NO_PREFETCH\nX=3 CCAMD_WAVE_BELOW=32\nX=4 CCAMD_WAVE_BELOW=48\n' | bash tools/env_sweep.sh
echo "### uniform frames"
printf 'X=1 -- --frame-kind uniform\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_WAVE_NO_PREFETCH -- --frame-kind uniform\n' | bash tools/env_sweep.sh
} > $O 2>&1
timeout -k 10 800 python -m pytest tests/test_gpu_specialize.py tests/test_gpu_detect.py -q -m gpu --timeout 300 > gpurun_out/r4_r_tests.log 2>&1
tail -n 3 gpurun_out/r4_r_tests.log >> $O
cat $O
