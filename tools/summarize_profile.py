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
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
stats = glob.glob(f"{src}/stats/*/*kernel_stats.csv")[0]
shutil.copy(stats, f"profiles/{dst_prefix}_kernel_stats.csv")
summary = {"kernels": {}}
for r in csv.DictReader(open(stats)):
    summary["kernels"][r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                                    "pct": float(r["Percentage"])}
for name, counter in (("pmc_rd", "FETCH_SIZE"), ("pmc_wr", "WRITE_SIZE")):
    f = glob.glob(f"{src}/{name}/*/*counter_collection.csv")
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
json.dump(summary, open(f"profiles/{dst_prefix}_summary.json", "w"), indent=1)
print(json.dumps(summary["kernels"], indent=1))
