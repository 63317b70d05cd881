#!/usr/bin/env python3
"""Gaps between consecutive cascade-kernel launches in a rocprofv3 kernel trace (csv): where a step's time goes that is not
cascade-kernel time. Usage: trace_gaps.py <dir with *_kernel_trace.csv> [kernel substring]"""
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
key = sys.argv[2] if len(sys.argv) > 2 else "k_eval"
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if key in r["Kernel_Name"])
ev = ev[len(ev) // 3:]  # skip warm-up
busy = sum(e - s for s, e in ev)
span = ev[-1][1] - ev[0][0]
gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
print("launches %d  span %.3f ms  kernel time %.3f ms (%.1f %%)  gaps: n %d  total %.3f ms  mean %.1f us  max %.1f us" %
      (len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, len(gaps), sum(gaps) / 1e6, sum(gaps) / len(gaps) / 1e3, max(gaps) / 1e3))
big = sorted(gaps)[-8:]
print("largest gaps (us):", [round(g / 1e3, 1) for g in big])
