#!/bin/bash
# Round 4, run W: batched negative mining (cc_negminer_run_batch) -- parity tests and throughput.
O=gpurun_out/r4_w.txt
{
timeout -k 10 600 python -m pytest tests/test_gpu_negmine.py tests/test_gpu_cpp_adaptor.py -q -m gpu --timeout 300 2>&1 | tail -n 15
echo "### tools/bench_negmine.py 10 5"
python tools/bench_negmine.py 10 5 2>/dev/null
} > $O 2>&1
cat $O
