#!/bin/bash
for wb in 8 16 24 32 48 64; do
  CCAMD_WAVE_BELOW=$wb python bench.py --cpu-frames 0 --specialize 7 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wave_below=$wb', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['eval_ms'])"
done
for p in 2 3 6 8; do
  CCAMD_PIPELINE_PASSES=$p python bench.py --cpu-frames 0 --specialize 7 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('passes=$p', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['eval_ms'])"
done
