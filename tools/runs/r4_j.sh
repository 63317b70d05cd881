#!/bin/bash
# Round 4, run J: the whole GPU suite on the final kernels, the committed evidence (tools/r4_final_profiles.sh), the driver's bench command.
timeout -k 10 1100 python -m pytest tests -v -m gpu --timeout 300 > gpurun_out/r4_t5.log 2>&1
tail -n 3 gpurun_out/r4_t5.log
bash tools/r4_final_profiles.sh
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err
tail -c 600 gpurun_out/r4_final_bench.json
