#!/bin/bash
# Round 4, run AB: the whole GPU suite and smoke() on the final tree.
timeout -k 10 1100 python -m pytest tests -v -m gpu --timeout 300 > gpurun_out/r4_t6.log 2>&1
tail -n 3 gpurun_out/r4_t6.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
