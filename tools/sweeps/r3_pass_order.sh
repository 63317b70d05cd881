for e in "X=1" "CCAMD_PASS_SIZES=7,19,19,19" "CCAMD_PASS_SIZES=4,20,20,20" "CCAMD_PASS_SIZES=2,21,21,20" "CCAMD_PASS_SIZES=7,19,19,12,7" "CCAMD_PASS_SIZES=4,15,15,15,15"; do
  env $e python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$e]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
