#!/bin/bash
# rocprofv3 kernel statistics of the front end for a cascade WITH tilted features (k_diag_sums, k_tilted_cols next to the
# resize / integral kernels): $2 (default 16) Full-HD frames per launch, device pipeline only. $1 = tag. Output: gpurun_out/tilted_<tag>/.
tag=${1:-t}
frames=${2:-16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tilted_$tag
mkdir -p $out
python3 - $out/tilted.xml <<'PY'
import sys
import numpy as np
from tests import cascade_factory as cf
from tests.util import frame_natural
img = frame_natural(320, 240, 3)
win = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
open(sys.argv[1], "w").write(cf.tilted_stump_cascade(win))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 bench.py --cascade $out/tilted.xml --specialize 0 --frames $frames --steps 3 --warmup 1 --cpu-frames 0 --device-only > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-600
python3 - $out/prof <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:10]:
    print("   %-60s calls %5s avg %10.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
