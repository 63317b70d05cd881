#!/bin/bash
# Sweep of a tuning knob (env var) over the full bench step (with copy-back and grouping).
knob=$1; shift
for v in "$@"; do
  env $knob=$v python bench.py --steps 5 --warmup 1 --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$knob=$v', 'Mwin/s', d['value'], 'step_ms', d['ms_per_step'], 'eval_ms', d['kernel_ms_per_step']['eval_ms'])"
done
