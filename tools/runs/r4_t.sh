#!/bin/bash
# Round 4, run T: one boosted LBP stage end to end (16 stumps, 20 000 samples, hard data), sorted-table search / streamed codes; 2 ranks on the one GPU over gloo.
O=gpurun_out/r4_t.txt
{
echo "### LBP stage, k_split_cat_sorted"
python tools/bench_boost_stage.py 16 20000 hard LBP 2>/dev/null | cut -c1-900
echo "### LBP stage, k_split_cat (CCAMD_SPLIT_CAT_STREAM=1)"
CCAMD_SPLIT_CAT_STREAM=1 python tools/bench_boost_stage.py 16 20000 hard LBP 2>/dev/null | cut -c1-900
echo "### bench.py, 2 ranks sharing the GPU, gloo"
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --no-extra --cpu-frames 0 2>gpurun_out/r4_t_2rank.err | cut -c1-700
} > $O 2>&1
cat $O
