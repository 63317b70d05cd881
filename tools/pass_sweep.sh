#!/bin/bash
# bench.py with explicit pipeline pass sizes (CCAMD_PASS_SIZES), 32 frames per step
for sz in "9,9,9,5" "2,4,8,13,5" "1,3,7,14,7" "2,5,10,11,4" "3,6,10,9,4" "2,4,8,10,8" "4,8,12,8"; do
  CCAMD_PASS_SIZES=$sz python bench.py --cpu-frames 0 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sizes=[$sz]', j['value'], j['ms_per_step'], j['kernel_ms_per_step']['eval_ms'])"
done
