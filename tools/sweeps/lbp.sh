for e in "X=1" "CCAMD_SPEC_PREFETCH=0" "CCAMD_SPEC_PREFETCH=2" "CCAMD_SPEC_BUDGET=640"; do
  env CCAMD_CACHE_DIR= $e python bench.py --cascade data/lbpcascade_frontalface.xml --specialize 20 --cpu-frames 0 --frames 32 --device-only --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', j['kernel_ms_per_step'])"
done
env python bench.py --cascade data/lbpcascade_frontalface.xml --specialize 0 --cpu-frames 0 --frames 32 --device-only --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('table-driven', j['kernel_ms_per_step'])"
