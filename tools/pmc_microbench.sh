#!/bin/bash
# Calibrates the LDS counters against known instruction streams: tools/microbench_cu.hip under rocprofv3 --pmc, one line
# per LDS variant with SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT per wavefront instruction next to the measured cycles.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_microbench
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o $out/microbench_cu tools/microbench_cu.hip || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $out/prof -- $out/microbench_cu > $out/microbench.txt 2>$out/err.txt
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/prof/*/*counter_collection.csv")[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_lds" not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
lines = [l.rstrip() for l in open(out + "/microbench.txt") if l.startswith("LDS ")]
# two launches per printed line; the second is the measured one
for i, l in enumerate(lines):
    d = disp[ids[2 * i + 1]]
    n = d.get("SQ_INSTS_LDS", 0) or 1
    print("%s | per LDS instr: IDX_ACTIVE %.2f BANK_CONFLICT %.2f (INSTS_LDS %.0f)" % (l, d.get("SQ_LDS_IDX_ACTIVE", 0) / n, d.get("SQ_LDS_BANK_CONFLICT", 0) / n, n)
          + "".join(" %s %.2f" % (k[7:], d[k] / n) for k in ("SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL") if k in d))
PY
