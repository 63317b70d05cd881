#!/bin/bash
# Issue / LDS / wait counters of the cascade kernel (two rocprofv3 passes of 8 SQ counters each, kernel trace only).
# $1 = tag, rest = bench args (e.g. --cascade data/lbpcascade_frontalface.xml --specialize 20). Writes gpurun_out/pmc_<tag>/summary.json
# (copy it to profiles/r04_pmc_eval.json or profiles/r04_pmc_eval_lbp.json: bench.py replays it while the kernel sources' hash matches).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CCAMD_BENCH_NO_VISITED=1 CCAMD_NO_FRONT_OVERLAP=1
out=gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out/a $out/b
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $out/a -- python3 bench.py --cpu-frames 0 --no-extra --steps 2 --warmup 1 "$@" > $out/log_a.txt 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAVES --output-format csv -d $out/b -- python3 bench.py --cpu-frames 0 --no-extra --steps 2 --warmup 1 "$@" > $out/log_b.txt 2>&1
python3 - "$out" "$@" <<'PY'
import csv, glob, collections, json, os, sys
out = sys.argv[1]
args = sys.argv[2:]
sys.path.insert(0, os.getcwd())
from bench import kernel_source_sha16
cascade = "haarcascade_frontalface_synthetic.xml"
for i, a in enumerate(args):
    if a == "--cascade":
        cascade = os.path.basename(args[i + 1])
def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
A, B = load(out + "/a"), load(out + "/b")
names = sorted([k for k in A if "k_eval" in k], reverse=True)  # k_eval_spec_step2 before k_eval_spec_step1
if not names:
    print("no k_eval kernel in the counter file; log tail:"); print(open(out + "/log_a.txt").read()[-1500:]); sys.exit(0)
# A pass is one launch of every cascade-kernel module (a specialised Haar kernel: one per step), and every module is launched
# equally often: per-pass totals = sums of the per-launch averages.
per = {}
m, mb = collections.defaultdict(float), collections.defaultdict(float)
for k in names:
    a = {c: sum(x) / len(x) for c, x in A[k].items()}
    b = {c: sum(x) / len(x) for c, x in B.get(k, {}).items()}
    per[k] = {"launches": len(next(iter(A[k].values()))), "counters_per_launch": {c: round(x) for c, x in a.items()},
              "counters_per_launch_pass_b": {c: round(x) for c, x in b.items()}}
    for c, x in a.items(): m[c] += x
    for c, x in b.items(): mb[c] += x
cus, simds = 256, 1024
cyc = m["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
o = {"kernel": "+".join(names), "cascade": cascade, "bench_args": args, "kernel_src_sha16": kernel_source_sha16(),
     "counters_per_pass": {c: round(x) for c, x in m.items()}, "counters_per_pass_b": {c: round(x) for c, x in mb.items()},
     "per_kernel": per,
     "kernel_cycles": round(cyc), "lds_pipeline_busy": round(m["SQ_LDS_IDX_ACTIVE"] / cus / cyc, 3),
     "lds_bank_conflict_share": round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3),
     "valu_busy": round(m["SQ_INSTS_VALU"] * 2.9 / simds / cyc, 3),
     "valu_active_counter_share": round(m["SQ_ACTIVE_INST_VALU"] * 4.0 / simds / cyc, 3)}
if mb.get("SQ_WAVE_CYCLES"):
    wc = mb["SQ_WAVE_CYCLES"]
    o["wave_cycles_waiting_on_counter_or_barrier"] = round(mb["SQ_WAIT_ANY"] / wc, 3)
    o["wave_cycles_ready_not_issued"] = round(mb["SQ_WAIT_INST_ANY"] / wc, 3)
    o["wave_cycles_issuing"] = round(mb["SQ_ACTIVE_INST_ANY"] / wc, 3)
    o["salu_insts_per_valu_inst"] = round(mb["SQ_INSTS_SALU"] / m["SQ_INSTS_VALU"], 3)
o["how"] = ("rocprofv3 --kernel-trace --pmc, two passes of 8 SQ counters, no stream overlap (CCAMD_NO_FRONT_OVERLAP=1: counters are device-wide), "
            "bench.py --cpu-frames 0 --no-extra --steps 2 --warmup 1 + bench_args; averages per launch, summed over the cascade kernel's modules "
            "(one launch each per pass). LDS busy = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs); conflict share = "
            "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; VALU busy = SQ_INSTS_VALU x 2.9 cycles / 1024 SIMDs / kernel cycles (2.9 = issue cost of the "
            "Haar kernel's instruction mix, tools/pmc_microbench.sh: add/sub/mov/mul 2.4-2.7, cmp/cndmask 3.8, cvt 4.2, f64 4.8 cycles per wavefront "
            "instruction on a saturated SIMD); valu_active_counter_share = the same counter at the nominal 4 cycles per instruction (upper bound); "
            "wave_cycles_*: SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES (disjoint shares of a wavefront's life)")
json.dump(o, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({x: o[x] for x in o if x not in ("how", "counters_per_pass", "counters_per_pass_b", "bench_args", "per_kernel")}))
PY
