#!/bin/bash
# Round 4, run V: random-cascade stress of the specialised kernels in this round's forms (tests/analysis/stress_specialised.py).
timeout -k 10 1000 python tests/analysis/stress_specialised.py ${1:-24} ${2:-0} > gpurun_out/r4_stress.txt 2>&1
echo "exit $?" >> gpurun_out/r4_stress.txt
tail -n 4 gpurun_out/r4_stress.txt
