#!/bin/bash
# bench.py with different specialisation depths (stages compiled into the cascade kernel)
for k in 0 2 4 5 6 7 9; do
  python bench.py --cpu-frames 0 --specialize $k 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$k', j['config']['kernel_specialized_stages'], j['value'], j['ms_per_step'], j['kernel_ms_per_step']['eval_ms'])"
done
