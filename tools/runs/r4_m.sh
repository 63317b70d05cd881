#!/bin/bash
# Round 4, run M: parity of the tile-height / module variants, wall time of the default bench command.
timeout -k 10 900 python -m pytest tests/test_gpu_specialize.py -x -v -m gpu --timeout 300 -k "tile_heights or queue_form" > gpurun_out/r4_t7.log 2>&1
tail -n 4 gpurun_out/r4_t7.log
/usr/bin/time -v python bench.py > gpurun_out/r4_default_bench.json 2> gpurun_out/r4_default_bench.err
grep -E "Elapsed|Maximum resident" gpurun_out/r4_default_bench.err
tail -c 300 gpurun_out/r4_default_bench.json
