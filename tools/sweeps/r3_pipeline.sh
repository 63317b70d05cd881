# Step time of bench.py under the knobs of the pass pipeline (run on the GPU box): pipelined submit / collect (default) and
# synchronous calls, pass counts and sizes, stream priority of the pyramid / integral stream.
for e in "X=1" "X=2" "CCAMD_EVEN_PASSES=1" "CCAMD_PIPELINE_PASSES=3 CCAMD_EVEN_PASSES=1" "CCAMD_PIPELINE_PASSES=2 CCAMD_EVEN_PASSES=1" "CCAMD_FRONT_SAME_PRIORITY=1"; do
  env $e python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$e]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra --sync-steps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[--sync-steps]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
