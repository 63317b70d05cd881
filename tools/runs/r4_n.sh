#!/bin/bash
# Round 4, run N: wall time of the default bench command; more compiled Haar stages with the module-per-step kernels.
O=gpurun_out/r4_n.txt
{
t0=$(date +%s.%N); python bench.py > gpurun_out/r4_default_bench.json 2> gpurun_out/r4_default_bench.err; t1=$(date +%s.%N)
echo "default bench.py wall seconds: $(echo "$t1 - $t0" | bc)"; tail -c 200 gpurun_out/r4_default_bench.json; echo
echo "### Haar: compiled stages (budget in stumps)"
printf 'X=1\nCCAMD_SPEC_BUDGET=420 -- --specialize 8\nCCAMD_SPEC_BUDGET=500 -- --specialize 9\nCCAMD_SPEC_BUDGET=600 -- --specialize 10\nCCAMD_SPEC_BUDGET=820 -- --specialize 12\nX=1 -- --specialize 6\nX=1 -- --specialize 5\n' | bash tools/env_sweep.sh
} > $O 2>&1
cat $O
