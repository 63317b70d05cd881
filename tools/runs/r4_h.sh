#!/bin/bash
# Round 4, run H: full GPU suite with one module per step (Haar) and 20-row LBP tiles, then the Haar module / tile-row sweep.
O=gpurun_out/r4_h.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
timeout -k 10 1100 python -m pytest tests/test_gpu_specialize.py tests/test_gpu_split.py tests/test_host_logic.py tests/test_host_sanitizers.py tests/test_oracle_detect.py tests/test_oracle_kats.py -x -v -m gpu --timeout 300 > gpurun_out/r4_t4.log 2>&1
tail -n 4 gpurun_out/r4_t4.log
grep -q " passed" gpurun_out/r4_t4.log && ! grep -q "failed\|Timeout" gpurun_out/r4_t4.log || exit 1
{
echo "### Haar: modules / tile rows (default = one module per step, 12 rows each)"
printf 'X=1\nCCAMD_SPEC_ONE_MODULE=1\nCCAMD_SPEC_TILE_Y1=8 CCAMD_SPEC_TILE_Y2=8\nCCAMD_SPEC_TILE_Y1=12 CCAMD_SPEC_TILE_Y2=8\nCCAMD_SPEC_TILE_Y1=16 CCAMD_SPEC_TILE_Y2=12\nCCAMD_SPEC_TILE_Y1=12 CCAMD_SPEC_TILE_Y2=16\nX=1 -- --content uniform\nCCAMD_SPEC_ONE_MODULE=1 -- --content uniform\n' | bash tools/env_sweep.sh
echo "### LBP default"
printf 'X=1 -- %s\n' "$LBP" | bash tools/env_sweep.sh
echo "### bench.py (driver command)"
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4_h_bench.err | tee gpurun_out/r4_h_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d[k] for k in ('value','value_synchronous','ms_per_step')}, 'host_frames', {k:d['host_frames'][k] for k in ('ms_per_step','value','synchronous_call_ms_per_step','rectangles_identical_to_resident_frames')}, 'eval avg', d['roofline']['avg_launch_ms'], d['roofline']['frac'])
print([ (w.get('workload')[:40], w.get('value'), w.get('cascade_kernel_ms_per_launch')) for w in d.get('extra_workloads',[])])"
tail -n 3 gpurun_out/r4_h_bench.err
} > $O 2>&1
tail -n 40 $O
