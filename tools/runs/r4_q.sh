#!/bin/bash
# run Q: where the categorical search's 1.8 ms go (kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out; O=gpurun_out/r4_q.txt; : > $O
out=gpurun_out/prof_cat; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_split_search.py LBP 20000 > $out/bench.log 2>&1 || { echo "PROFILE FAILED" >> $O; tail -5 $out/bench.log >> $O; exit 1; }
python3 - $out >> $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print("   %-70s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $out -name "*.csv" ! -name "*kernel_stats.csv" -delete
