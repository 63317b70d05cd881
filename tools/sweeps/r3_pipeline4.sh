# Pipelined bench steps: the cascade kernel compiled for 7 wavefronts per SIMD (72 VGPRs) against the default budget, twice each
for e in "CCAMD_SPEC_WAVES_PER_EU=7" "X=1" "CCAMD_SPEC_WAVES_PER_EU=7" "X=2"; do
  env $e python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$e]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
