#!/bin/bash
# One rocprofv3 counter pass over bench.py (device pipeline only, 32 frames per launch) for the k_eval_* kernel.
# $1 = tag, $2 = quoted counter list, rest = extra bench args. Prints the per-launch averages.
tag=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CCAMD_BENCH_NO_VISITED=1 CCAMD_NO_FRONT_OVERLAP=1
out=gpurun_out/pmc_any_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 bench.py --cpu-frames 0 --steps 2 --warmup 1 --frames 32 --device-only "$@" > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/*/*counter_collection.csv")
if not f:
    print("no counters; log tail:"); print(open(out + "/log.txt").read()[-1500:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "k_eval" in k:
        print(k, " ".join("%s=%.2fM" % (c, sum(x) / len(x) / 1e6) for c, x in sorted(v.items())))
PY
