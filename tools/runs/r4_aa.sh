#!/bin/bash
# Round 4, run AA: pass sizes of the SYNCHRONOUS call (cc_detect_batch of 64 frames; bench --sync-steps), CCAMD_PASS_SIZES.
O=gpurun_out/r4_aa.txt
{
for ps in "" "19,19,19,7" "4,12,20,20,8" "6,16,22,14,6" "8,24,24,8" "4,8,16,28,8" "12,22,22,8" "16,20,20,8"; do
CCAMD_PASS_SIZES=$ps python bench.py --steps 20 --warmup 5 --sync-steps --no-extra --cpu-frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('sizes [%s]' % '$ps', 'ms_per_step', d['ms_per_step'], 'value', d['value'])"
done
} > $O 2>&1
cat $O
