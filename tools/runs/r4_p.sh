#!/bin/bash
# run P: categorical split search from the sorted table
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/r4_p.txt; : > $O
echo "### split tests" >> $O
timeout -k 10 500 python -m pytest tests/test_gpu_split.py tests/test_gpu_config_sizes.py -m gpu -x -q --timeout 400 >> $O 2>&1 || { echo "TESTS FAILED" >> $O; exit 1; }
echo "### LBP split search: sorted table / streamed codes" >> $O
timeout -k 10 300 python tools/bench_split_search.py LBP 20000 >> $O 2>&1 || exit 1
CCAMD_SPLIT_CAT_STREAM=1 timeout -k 10 300 python tools/bench_split_search.py LBP 20000 >> $O 2>&1 || exit 1
for P in 1 4 14; do echo "parts=$P" >> $O; CCAMD_SPLIT_CAT_PARTS=$P timeout -k 10 300 python tools/bench_split_search.py LBP 20000 2>/dev/null | cut -c1-560 >> $O || exit 1; done
