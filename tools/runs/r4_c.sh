#!/bin/bash
# Round 4, run C: tests of the changed paths, split search lean vs branchy, unedited trainer loop, default bench line.
O=gpurun_out/r4_c.txt
{
echo "### tests"
timeout -k 10 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_cpp_adaptor.py tests/test_gpu_detect.py tests/test_gpu_split.py tests/test_gpu_negmine.py -x -q -m gpu 2>&1 | tail -n 15
echo "### split search lean / branchy"
python tools/bench_split_search.py HAAR 20000 2>/dev/null
CCAMD_SPLIT_BRANCHY=1 python tools/bench_split_search.py HAAR 20000 2>/dev/null
echo "### unedited trainer loop"
cascadeclassifier_amd/lib/bench_unedited_trainer data/haarcascade_frontalface_synthetic.xml 10
echo "### bench.py (driver command)"
python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4_c_bench.err | tee gpurun_out/r4_c_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d[k] for k in ('value','value_synchronous','ms_per_step')}, 'host_frames', d['host_frames'], 'eval avg', d['roofline']['avg_launch_ms'])
print([ (w.get('name') or w.get('workload'), w.get('value'), w.get('cascade_kernel_ms_per_launch')) for w in d.get('extra_workloads',[])])"
tail -n 5 gpurun_out/r4_c_bench.err
} > $O 2>&1
tail -n 40 $O
