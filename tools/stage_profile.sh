#!/bin/bash
# Kernel time of the cascade evaluation as a function of cascade depth (device pipeline only).
set -e
for n in 1 2 3 4 6 8 12 25; do
  python tools/truncate_cascade.py data/haarcascade_frontalface_synthetic.xml $n /tmp/casc_$n.xml
  python bench.py --steps 3 --warmup 1 --cpu-frames 0 --device-only --cascade /tmp/casc_$n.xml 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('stages', $n, 'eval_ms/16f', d['kernel_ms_per_step']['eval_ms'], 'step_ms', d['ms_per_step'])"
done
