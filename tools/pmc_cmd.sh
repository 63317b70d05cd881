#!/bin/bash
# One rocprofv3 counter pass (kernel trace + --pmc only) over an arbitrary python tool; prints per-launch averages of the
# kernels whose name contains $3. $1 = tag, $2 = quoted counter list, $3 = kernel-name filter, rest = python script + args.
tag=$1; ctr=$2; filt=$3; shift; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_cmd_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 "$@" > $out/log.txt 2>&1
python3 - $out "$filt" <<'PY'
import csv, glob, collections, sys
out, filt = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/*/*counter_collection.csv")
if not f:
    print("no counters; log tail:"); print(open(out + "/log.txt").read()[-1500:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if filt in k:
        print(k[:60], "launches", len(next(iter(v.values()))), " ".join("%s=%.3fM" % (c, sum(x) / len(x) / 1e6) for c, x in sorted(v.items())))
PY
