# Pass counts for pipelined (submit / collect) steps of bench.py: with the next batch overlapping the last pass, even passes.
for e in "CCAMD_PIPELINE_PASSES=1" "CCAMD_PIPELINE_PASSES=2 CCAMD_EVEN_PASSES=1" "CCAMD_PIPELINE_PASSES=2" "CCAMD_PIPELINE_PASSES=4 CCAMD_EVEN_PASSES=1"; do
  env $e python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$e]', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
