#!/bin/bash
# Runs the device-only bench (32 frames per launch) once per line of stdin; a line is a list of VAR=value settings
# followed by optional bench arguments after "--".
while read -r line; do
  [ -z "$line" ] && continue
  envs="${line%%--*}"; args=""
  case "$line" in *--*) args="${line#*--}";; esac
  env CCAMD_CACHE_DIR=/tmp/ccamd_sweep_cache $envs python bench.py --steps 3 --warmup 1 --cpu-frames 0 --frames 32 --device-only $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('''$line''', '=> eval_ms/32f', d['kernel_ms_per_step']['eval_ms'], 'spec', d['config']['kernel_specialized_stages'])"
done
