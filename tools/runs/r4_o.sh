#!/bin/bash
# Round 4, run O: smoke(), then knobs re-checked on the module-per-step Haar kernel (wave-phase threshold, list queue, passes per batch).
O=gpurun_out/r4_o.txt
{
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
echo "### Haar (12 / 8 rows, two modules): wave-phase threshold, list queue"
printf 'X=1\nCCAMD_WAVE_BELOW=16\nCCAMD_WAVE_BELOW=32\nCCAMD_WAVE_BELOW=48\nCCAMD_DENSE_FROM=4\nCCAMD_DENSE_FROM=5\nCCAMD_DENSE_FROM=6\nCCAMD_SPLIT_STUMPS=0\n' | bash tools/env_sweep.sh
echo "### pipelined bench step: passes per submitted batch"
for p in 1 2 3 4; do CCAMD_PIPELINE_PASSES=$p python bench.py --steps 20 --warmup 5 --no-extra --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('passes', $p, 'ms_per_step', d['ms_per_step'], 'value', d['value'])"; done
} > $O 2>&1
cat $O
