#!/usr/bin/env python3
"""Per-phase durations of the cascade kernel's blocks from a CCAMD_DEBUG_STAMPS dump (shader-clock cycles of wavefront 0)."""
import sys
import numpy as np
raw = open(sys.argv[1], "rb").read()
n_tiles, nf, slots, _ = np.frombuffer(raw[:16], np.int32)
st = np.frombuffer(raw[16:], np.uint64).reshape(-1, slots).astype(np.int64)
st = st[st[:, 0] > 0]
end = st[:, slots - 1]
ok = end > 0
st, end = st[ok], end[ok]
total = end - st[:, 0]
print("blocks", len(st), "mean block cycles", total.mean(), "median", np.median(total))
prev = st[:, 0].copy()
names = ["staging", "dense(var+s0)"] + ["stage %d" % k for k in range(1, slots - 3)]
if slots == 40:  # wave-phase stamps: slot 30 = windows collected, 31 + i = i-th wave-phase stage done
    names[29] = "W: collect"
    for i in range(8):
        names[30 + i] = "W: stage +%d" % i
for k in range(1, slots - 1):
    cur = st[:, k]
    have = cur > 0
    if have.sum() == 0:
        continue
    d = np.where(have, cur - prev, 0)
    print("%-14s reached by %5.1f%% of blocks  mean cycles (over all blocks) %8.0f  (over blocks that ran it) %8.0f" % (names[k - 1], 100 * have.mean(), d.mean(), d[have].mean()))
    prev = np.where(have, cur, prev)
tail = end - prev
print("%-14s mean cycles %8.0f" % ("rest (wave phase tail)", tail.mean()))
k0 = st[:, 0].min()
span = end.max() - k0
print("kernel span cycles", span, "sum of block cycles / span =", total.sum() / span, "(average blocks in flight)")
