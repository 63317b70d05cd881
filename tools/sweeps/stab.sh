for i in 1 2 3 4 5 6; do
  timeout -k 5 120 python bench.py --steps 5 --warmup 1 --cpu-frames 0 --frames 32 --device-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('run $i eval_ms/32f', d['kernel_ms_per_step']['eval_ms'])"
done
