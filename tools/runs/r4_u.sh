#!/bin/bash
# Round 4, run U: wall time of the default bench command (no flags) and of smoke(); counter replay accepted for the committed sources.
O=gpurun_out/r4_u.txt
{
SECONDS=0
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
echo "smoke(): $SECONDS s"
SECONDS=0
python bench.py > gpurun_out/r4_u_bench.json 2> gpurun_out/r4_u_bench.err
echo "python bench.py (defaults): $SECONDS s, exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_u_bench.json").readline())
r = d["roofline"]
print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "steps", d["steps"], "warmup", d["warmup"])
print("roofline frac", r["frac"], "traffic", r["traffic"], "traffic_source", r.get("traffic_source"))
print("secondary", r.get("secondary"))
print("cpu_baseline", d["cpu_baseline"])
for w in d.get("extra_workloads", []):
    print(" extra:", w["workload"][:70], w.get("value"), w.get("traffic"))
PY
} > $O 2>&1
cat $O
