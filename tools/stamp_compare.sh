#!/bin/bash
# Per-phase block stamps of the cascade kernel for a list of environment settings (one per line on stdin), 8 frames,
# device-only. Each line: VAR=value ... [-- bench args]. Output: tools/analyze_stamps.py per setting.
while read -r line; do
  [ -z "$line" ] && continue
  envs="${line%%--*}"; args=""
  case "$line" in *--*) args="${line#*--}";; esac
  echo "=== $line"
  rm -f /tmp/stamps.bin
  env CCAMD_CACHE_DIR=/tmp/ccamd_sweep_cache CCAMD_NO_FRONT_OVERLAP=1 CCAMD_DEBUG_STAMPS=/tmp/stamps.bin CCAMD_PIPELINE_PASSES=1 $envs python bench.py --steps 1 --warmup 1 --cpu-frames 0 --frames 8 --device-only $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('eval_ms/8f', d['kernel_ms_per_step']['eval_ms'])"
  python tools/analyze_stamps.py /tmp/stamps.bin
done
