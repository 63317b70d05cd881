#!/bin/bash
# Round 4, run E: LBP kernel -- issue priority knobs, tile shapes; host frames after the 3-slot staging; tests of the staging.
O=gpurun_out/r4_e.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
echo "### tests (detect: staging slots)"
timeout -k 10 600 python -m pytest tests/test_gpu_detect.py -x -q -m gpu 2>&1 | tail -n 3
echo "### LBP: issue priority while staging / after the dense phase"
printf 'X=1 -- %s\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_PRIO_STAGE=3 -- %s\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_PRIO_STAGE=1 -- %s\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_PRIO_LATE=1 -- %s\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_PRIO_LATE=3 -- %s\n' "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" | bash tools/env_sweep.sh
echo "### Haar: the same"
printf 'X=1\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_PRIO_STAGE=3\nCCAMD_SPEC_EXTRA_FLAGS=-DCC_PRIO_LATE=2\n' | bash tools/env_sweep.sh
echo "### LBP tile shapes (window rows x threads)"
for v in 16_256 16_512 12_256 4_256 8_512; do
  for w in 7 5; do
    CCAMD_LIB=cascadeclassifier_amd/lib/libcascadeclassifier_amd_$v.so CCAMD_CACHE_DIR= CCAMD_TRACE_HOST=1 CCAMD_SPEC_WAVES_PER_EU=$w python bench.py --steps 3 --warmup 1 --cpu-frames 0 --frames 32 --device-only $LBP 2> >(grep "resident blocks" >&2) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v', 'waves_per_eu', $w, 'eval_ms/32f', d['kernel_ms_per_step']['eval_ms'])"
  done
done
echo "### bench.py (driver command): host frames"
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4_e_bench.err | tee gpurun_out/r4_e_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d[k] for k in ('value','value_synchronous','ms_per_step')}, 'host_frames', {k:d['host_frames'][k] for k in ('ms_per_step','value','synchronous_call_ms_per_step','rectangles_identical_to_resident_frames')})"
} > $O 2>&1
tail -n 50 $O
