#!/bin/bash
# Per-stage cost of the cascade kernel with counters: for each stop point (CCAMD_DEBUG_STOP_AFTER_STAGE: -2 = tile staging
# only, -3 = + variance test, k = through stage k) one rocprofv3 pass with the LDS / VALU counters of k_eval_*.
# $1 = tag, $2 = --specialize value, rest = stop points (default list below). Output: gpurun_out/stage_pmc_<tag>.txt
tag=$1; spec=${2:-7}; shift; shift
stops=${@:--2 -3 0 1 2 3 4 5 6 8 24}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CCAMD_BENCH_NO_VISITED=1 CCAMD_NO_FRONT_OVERLAP=1
res=gpurun_out/stage_pmc_$tag.txt
: > $res
for s in $stops; do
  out=gpurun_out/stage_pmc_$tag/s$s
  mkdir -p $out
  CCAMD_DEBUG_STOP_AFTER_STAGE=$s rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $out -- python3 bench.py --cpu-frames 0 --steps 2 --warmup 1 --frames 32 --device-only --specialize $spec > $out/log.txt 2>&1
  python3 - $s $out >> $res <<'PY'
import csv, glob, collections, sys
s, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/*/*counter_collection.csv")
if not f:
    print("stop", s, "no counters"); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "k_eval" in k:
        m = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        print("stop", s, k, "kernel_Mcycles %.2f" % (cyc / 1e6), " ".join("%s=%.1fM" % (c, x / 1e6) for c, x in sorted(m.items()) if c != "GRBM_GUI_ACTIVE"))
PY
  tail -1 $res
done
