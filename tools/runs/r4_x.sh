#!/bin/bash
# Round 4, run X: evaluator / split / negmine / concurrency tests after the copies left the legacy stream.
timeout -k 10 900 python -m pytest tests/test_gpu_random_parity.py tests/test_gpu_eval.py tests/test_gpu_split.py tests/test_gpu_negmine.py tests/test_gpu_cpp_adaptor.py -q -m gpu --timeout 400 > gpurun_out/r4_x.log 2>&1
tail -n 5 gpurun_out/r4_x.log
