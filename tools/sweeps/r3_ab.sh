# A/B of two library builds on one box: pipelined bench steps, synchronous steps and single-image latency, alternating
L=cascadeclassifier_amd/lib
for rep in 1 2 3 4; do
  for lib in libcascadeclassifier_amd_old.so libcascadeclassifier_amd.so; do
    CCAMD_LIB=$L/$lib python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$lib pipelined]', d['ms_per_step'], d['kernel_ms_per_step']['eval_ms'], d['kernel_ms_per_step']['integral_ms'])"
  done
done
for lib in libcascadeclassifier_amd_old.so libcascadeclassifier_amd.so; do
  CCAMD_LIB=$L/$lib python bench.py --steps 10 --warmup 3 --cpu-frames 0 --no-extra --sync-steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$lib sync]', d['ms_per_step'])"
  CCAMD_LIB=$L/$lib python tools/bench_latency.py 2>/dev/null | tail -1
done
