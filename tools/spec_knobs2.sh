#!/bin/bash
# Sweep of the specialised kernel's code-generation knobs (prefetch depth x register budget); device pipeline only.
for d in ${DEPTHS:-1 2 3}; do for w in ${WAVES:-5 6}; do
  CCAMD_CACHE_DIR= CCAMD_SPEC_PREFETCH=$d CCAMD_SPEC_WAVES_PER_EU=$w python bench.py --steps 3 --warmup 1 --cpu-frames 0 --frames 32 --device-only --specialize ${1:-7} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('prefetch', $d, 'waves_per_eu', $w, 'eval_ms/32f', d['kernel_ms_per_step']['eval_ms'])"
done; done
