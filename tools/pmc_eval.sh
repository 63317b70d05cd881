#!/bin/bash
# Issue / LDS counters of the cascade kernel (own rocprofv3 pass, kernel trace only). $1 = tag, rest = bench args.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export CCAMD_BENCH_NO_VISITED=1 CCAMD_NO_FRONT_OVERLAP=1
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $out -- python3 bench.py --cpu-frames 0 --no-extra --steps 2 --warmup 1 "$@" > $out/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "k_eval" in k:
        m={c:sum(x)/len(x) for c,x in v.items()}
        print(k, {c:round(x) for c,x in m.items()})
        if m.get("SQ_LDS_IDX_ACTIVE"):
            print("  LDS conflict share of LDS-active cycles:", round(m["SQ_LDS_BANK_CONFLICT"]/m["SQ_LDS_IDX_ACTIVE"],3))
        import json
        cus, simds = 256, 1024
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        import os, sys
        sys.path.insert(0, os.getcwd())
        from bench import kernel_source_sha16
        out = {"kernel": k, "kernel_src_sha16": kernel_source_sha16(), "counters_per_launch": {c: round(x) for c, x in m.items()},
               "kernel_cycles": round(cyc), "lds_pipeline_busy": round(m["SQ_LDS_IDX_ACTIVE"] / cus / cyc, 3),
               "lds_bank_conflict_share": round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 3),
               "valu_busy": round(m["SQ_INSTS_VALU"] * 2.9 / simds / cyc, 3),
               "valu_active_counter_share": round(m["SQ_ACTIVE_INST_VALU"] * 4.0 / simds / cyc, 3),
               "how": "rocprofv3 --kernel-trace --pmc (one pass, no stream overlap), bench.py --cpu-frames 0 --steps 2 --warmup 1; LDS busy = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs); VALU busy = SQ_INSTS_VALU x 2.9 cycles / 1024 SIMDs / kernel cycles (2.9 = issue cost of the kernel's instruction mix from tools/pmc_microbench.sh: add/sub/mov/mul 2.4-2.7, cmp/cndmask 3.8, cvt 4.2, f64 4.8 cycles per wavefront instruction on a saturated SIMD; the nominal 4 cycles would read 1.48 on a saturated v_add_u32 stream); valu_active_counter_share = SQ_ACTIVE_INST_VALU (quad-cycles, one per issued instruction on this part) x 4 / 1024 SIMDs / kernel cycles: the nominal-rate reading of the same counter, an upper bound"}
        json.dump(out, open("$out/summary.json", "w"), indent=1)
        print("  LDS pipeline busy:", out["lds_pipeline_busy"], " VALU busy:", out["valu_busy"])
        if m.get("SQ_BUSY_CYCLES"):
            print("  LDS active / SQ busy:", round(m["SQ_LDS_IDX_ACTIVE"]/m["SQ_BUSY_CYCLES"],3), " VALU active / SQ busy:", round(m["SQ_ACTIVE_INST_VALU"]/m["SQ_BUSY_CYCLES"],3))
PY
