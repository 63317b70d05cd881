#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the small summaries committed under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, dst_prefix = sys.argv[1], sys.argv[2]
suffix = sys.argv[3] if len(sys.argv) > 3 else ""  # "_lbp": the traffic file of another cascade's kernel (profiles/<round>_traffic_k_eval_lbp.json)
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
newest = lambda pattern: sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]  # a tag may have been profiled more than once
stats = newest(f"{src}/stats/*/*kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{dst_prefix}_kernel_stats.csv")
summary = {"kernels": {}}
for r in csv.DictReader(open(stats)):
    summary["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                    "pct": float(r["Percentage"])}
for name, counter in (("pmc_rd", "FETCH_SIZE"), ("pmc_wr", "WRITE_SIZE")):
    f = newest(f"{src}/{name}/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summary["kernels"].setdefault(k, {})[counter + "_KB_per_launch_raw"] = sum(v) / len(v)
try:
    summary["bench"] = json.loads(open(f"{src}/bench_plain.json").read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    summary["bench_error"] = str(e)
# HBM traffic of the cascade kernel, with the identity of the kernel sources it was measured on (bench.py replays the
# figure only for the same sources). FETCH_SIZE counts 64 B per 128-B request on gfx950 (x2), WRITE_SIZE is exact.
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_sha16  # noqa: E402
summary["kernel_src_sha16"] = kernel_source_sha16()
# a pass is one launch of every cascade-kernel module (a specialised Haar kernel: one per step): sum their per-launch averages
ev = sorted([k for k, v in summary["kernels"].items() if k.startswith("k_eval") and "FETCH_SIZE_KB_per_launch_raw" in v and "WRITE_SIZE_KB_per_launch_raw" in v],
            reverse=True)
if ev and "bench" in summary:
    fpl = summary["bench"]["roofline"]["frames_per_launch"]
    fetch = sum(summary["kernels"][k]["FETCH_SIZE_KB_per_launch_raw"] for k in ev)
    write = sum(summary["kernels"][k]["WRITE_SIZE_KB_per_launch_raw"] for k in ev)
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_bench.sh {tag}), MI355X, bench.py "
                         "--cpu-frames 0 --steps 2 --warmup 1, CCAMD_NO_FRONT_OVERLAP=1 (counters are device-wide); per pass = summed over the "
                         "cascade kernel's modules",
               "kernel": "+".join(ev), "cascade": summary["bench"]["config"]["cascade"].split(" ")[0], "kernel_src_sha16": summary["kernel_src_sha16"],
               "frames_per_launch": fpl, "fetch_correction": 2.0,
               "fetch_size_kb_raw_per_launch": fetch, "write_size_kb_raw_per_launch": write,
               "per_kernel_kb_raw": {k: {"fetch": summary["kernels"][k]["FETCH_SIZE_KB_per_launch_raw"], "write": summary["kernels"][k]["WRITE_SIZE_KB_per_launch_raw"]} for k in ev},
               "hbm_bytes_per_frame": (2.0 * fetch + write) * 1024 / fpl},
              open(f"profiles/{dst_prefix.split('_')[0]}_traffic_k_eval{suffix}.json", "w"), indent=1)
json.dump(summary, open(f"profiles/{dst_prefix}_summary.json", "w"), indent=1)
print(json.dumps(summary["kernels"], indent=1))
