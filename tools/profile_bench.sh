#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box into gpurun_out/prof_<tag>/ :
#   stats/   --kernel-trace --stats          (per-kernel average durations)
#   pmc_rd/  --pmc FETCH_SIZE                (HBM-side read traffic; own pass, see MI355X_MICROARCH.md §HBM)
#   pmc_wr/  --pmc WRITE_SIZE
# Usage: tools/profile_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
# only the headline's timed launches in the trace: no visited-window launch (one extra 1-frame launch), none of the legs bench.py
# reports beside the headline (--no-extra: host split, host frames, LBP and uniform-noise workloads use the same kernels)
export CCAMD_BENCH_NO_VISITED=1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --cpu-frames 0 --no-extra "$@" > $out/bench_stats.log 2>&1
# PMC counters are device-wide while a kernel runs: keep the pyramid/integral stream from overlapping the cascade kernel
export CCAMD_NO_FRONT_OVERLAP=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_rd -- python3 bench.py --cpu-frames 0 --no-extra --steps 2 --warmup 1 "$@" > $out/bench_pmc_rd.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_wr -- python3 bench.py --cpu-frames 0 --no-extra --steps 2 --warmup 1 "$@" > $out/bench_pmc_wr.log 2>&1
unset CCAMD_NO_FRONT_OVERLAP CCAMD_BENCH_NO_VISITED
python3 bench.py "$@" > $out/bench_plain.json 2> $out/bench_plain.err
tail -1 $out/bench_plain.json
