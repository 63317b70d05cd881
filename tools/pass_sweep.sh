#!/bin/bash
# bench.py (64 frames per step) with explicit pipeline pass sizes (CCAMD_PASS_SIZES)
for sz in "19,19,19,7" "8,20,20,16" "6,18,18,16,6" "10,22,22,10" "12,15,15,15,7" "16,16,16,16"; do
  CCAMD_PASS_SIZES=$sz python bench.py --cpu-frames 0 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sizes=[$sz]', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['eval_ms'])"
done
