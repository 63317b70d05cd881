#!/bin/bash
# Round 4, run I: defaults check (Haar 12 / 8 rows in two modules, LBP 20 rows), LBP with a module per step, the bulk
# evaluator's store stream in 256- and 512-byte pieces.
O=gpurun_out/r4_i.txt
LBP="--cascade data/lbpcascade_frontalface.xml --specialize 20"
{
echo "### Haar default, LBP default and LBP with one module per step"
printf 'X=1\nX=1 -- %s\nCCAMD_SPEC_TWO_MODULES=1 -- %s\nCCAMD_SPEC_TWO_MODULES=1 CCAMD_SPEC_TILE_Y1=12 -- %s\nCCAMD_SPEC_TWO_MODULES=1 CCAMD_SPEC_TILE_Y1=16 -- %s\nCCAMD_SPEC_TWO_MODULES=1 CCAMD_SPEC_TILE_Y1=24 CCAMD_SPEC_TILE_Y2=20 -- %s\n' "$LBP" "$LBP" "$LBP" "$LBP" "$LBP" | bash tools/env_sweep.sh
echo "### bulk evaluator: whole kernel / arithmetic without stores / stores alone in 256-B pieces / in 512-B pieces"
python tools/bench_training_eval.py 2>/dev/null | tail -n 2
CCAMD_DEBUG_EVAL_NOSTORE=1 python tools/bench_training_eval.py 2>/dev/null | tail -n 1
CCAMD_DEBUG_EVAL_NOSTORE=2 python tools/bench_training_eval.py 2>/dev/null | tail -n 1
CCAMD_DEBUG_EVAL_NOSTORE=3 python tools/bench_training_eval.py 2>/dev/null | tail -n 1
} > $O 2>&1
tail -n 30 $O
