#!/bin/bash
# Cascade-kernel time as a function of where the kernel is told to stop (CCAMD_DEBUG_STOP_AFTER_STAGE: -2 = tile staging
# only, -3 = + variance test, k = through stage k), device pipeline only, 32 frames per step. $1 = --specialize value.
for s in -2 -3 0 1 2 3 4 5 6 7 8 10 14 24; do
  CCAMD_NO_FRONT_OVERLAP=1 CCAMD_DEBUG_STOP_AFTER_STAGE=$s python bench.py --steps 3 --warmup 1 --cpu-frames 0 --frames 32 --device-only --specialize ${1:-7} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('stop_after', $s, 'eval_ms/32f', d['kernel_ms_per_step']['eval_ms'])"
done
