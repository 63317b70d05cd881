for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 10 --warmup 2" "--steps 50 --warmup 5" "--steps 100 --warmup 5" "--steps 20 --warmup 5"; do
  python bench.py --gpus 1 $a --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$a', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['eval_ms'], d['roofline']['avg_launch_ms'])"
done
