#!/bin/bash
# Round 4, run D: blocked lean split search, trainer loop (window loop only), LBP LDS-vs-VALU sensitivity.
O=gpurun_out/r4_d.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
echo "### split search: tests + lean (blocks of 4 ranks) / branchy"
timeout -k 10 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -n 3
python tools/bench_split_search.py HAAR 20000 2>/dev/null
CCAMD_SPLIT_BRANCHY=1 python tools/bench_split_search.py HAAR 20000 2>/dev/null
echo "### unedited trainer loop (10 trained stages)"
python tools/truncate_cascade.py data/haarcascade_frontalface_synthetic.xml 10 /tmp/trunc10.xml
cascadeclassifier_amd/lib/bench_unedited_trainer /tmp/trunc10.xml
cascadeclassifier_amd/lib/bench_unedited_trainer /tmp/trunc10.xml 10 1920 1080
echo "### LBP sensitivity: 3 = every corner read twice, 4 = stump arithmetic twice"
printf 'X=1 -- %s\nCCAMD_DEBUG_SPEC_MODE=3 -- %s\nCCAMD_DEBUG_SPEC_MODE=4 -- %s\nCCAMD_DEBUG_SPEC_MODE=4 CCAMD_SPEC_WAVES_PER_EU=5 -- %s\nCCAMD_SPEC_WAVES_PER_EU=5 -- %s\n' "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" | bash tools/env_sweep.sh
} > $O 2>&1
tail -n 40 $O
