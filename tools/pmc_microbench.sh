#!/bin/bash
# Calibrates the LDS counters against known instruction streams: tools/microbench_cu.hip under rocprofv3 --pmc, one line
# per LDS variant with SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT per wavefront instruction next to the measured cycles.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_microbench
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o $out/microbench_cu tools/microbench_cu.hip || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL --output-format csv -d $out/prof -- $out/microbench_cu > $out/microbench.txt 2>$out/err.txt
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/prof_valu -- $out/microbench_cu > $out/microbench_valu.txt 2>$out/err_valu.txt
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
# VALU streams: counters per wavefront instruction next to the measured issue cycles
f = glob.glob(out + "/prof_valu/*/*counter_collection.csv")[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_valu" not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
lines = [l.rstrip() for l in open(out + "/microbench_valu.txt") if l.startswith("VALU ")]
per = len(ids) // max(len(lines), 1)  # launches per printed line; the last one is the measured one
for i, l in enumerate(lines):
    d = disp[ids[per * i + per - 1]]
    n = d.get("SQ_INSTS_VALU", 0) or 1
    print("%s | per VALU instr: ACTIVE_INST_VALU %.2f ; ACTIVE_INST_VALU*4 / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs) = %.3f ; INSTS_VALU*4 / same = %.3f" % (
        l, d.get("SQ_ACTIVE_INST_VALU", 0) / n, d.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (d["GRBM_GUI_ACTIVE"] / 8 * 1024), n * 4 / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)))
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
