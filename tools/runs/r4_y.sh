#!/bin/bash
# Round 4, run Y: the default bench command with the two new training-side extra workloads; wall time.
SECONDS=0
python bench.py > gpurun_out/r4_y_bench.json 2> gpurun_out/r4_y_bench.err
echo "python bench.py (defaults): $SECONDS s, exit $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_y_bench.json").readline())
print("value", d["value"], "ms_per_step", d["ms_per_step"], "traffic", d["roofline"]["traffic"])
for w in d.get("extra_workloads", []):
    print(" extra:", json.dumps(w)[:420])
PY
tail -n 3 gpurun_out/r4_y_bench.err
