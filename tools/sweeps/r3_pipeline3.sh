# Register budget of the specialised cascade kernel against the front end sharing the CUs (pipelined bench steps): with fewer
# registers per cascade wavefront, wavefronts of the integral kernels (99-121 VGPRs) fit beside five cascade wavefronts per SIMD.
for e in "X=1" "CCAMD_SPEC_WAVES_PER_EU=6" "CCAMD_SPEC_WAVES_PER_EU=7" "X=2"; do
  env $e python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$e]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
