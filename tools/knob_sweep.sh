#!/bin/bash
# Sweep of a tuning knob (env var) over the device-only bench.
knob=$1; shift
for v in "$@"; do
  env $knob=$v python bench.py --steps 3 --warmup 1 --cpu-frames 0 --device-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$knob=$v', 'eval_ms/16f', d['kernel_ms_per_step']['eval_ms'], 'step_ms', d['ms_per_step'])"
done
