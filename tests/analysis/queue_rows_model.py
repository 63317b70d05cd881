#!/usr/bin/env python3
"""What the thread-per-window phase of the cascade kernel pays for, computed from the CPU oracle's per-window results on one
bench frame (1920x1080 natural-like, stock-profile Haar cascade, scaleFactor 1.1; STEP-2 levels only).

For every 64 x 8-window tile and every stage k >= 1: the windows that reach the stage, the rows R of the bank-class queue table
(= the fullest of the 32 classes = the LDS cycles a corner gather of the stage costs, however the windows are packed), the
passes the block's 4 wavefronts make (whole rounds of 8 rows + the leftover row groups split by stumps), and the same for
the pair tile's slots (one or two neighbouring windows; greedy pairing along each row, re-paired at every stage = a lower bound
for the kernel, which only splits pairs). Printed means are over the tiles that still hold a window.

Analysis helper, not a test (it lives here because it uses the oracle): python tests/analysis/queue_rows_model.py
Output kept in profiles/r03_queue_rows_model.txt; DESIGN.md 4.4.1 quotes it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests.util import frame_natural  # noqa: E402


def passes(rows, max_slices):
    groups = (rows + 1) // 2          # a wavefront takes two rows (one per half)
    full = groups - groups % 4        # whole rounds of the 4 wavefronts
    rem = groups - full
    ns = 1
    if rem > 0:
        while ns * 2 * rem <= 4 and ns * 2 <= max_slices:
            ns *= 2
    return full / 4 + (1.0 / ns if rem > 0 else 0.0)


def main():
    c = O.load_cascade_xml(os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"))
    img = frame_natural(1920, 1080, 5)
    rd = O.detect_raw(c, img, 1.1, nthreads=8, full=True)
    sc = O.scales(c.win_w, c.win_h, 1920, 1080, 1.1)
    K = 8
    stat = {k: dict(tiles=0, n=0, slots=0, r1=0, rp=0, p1=0.0, pp=0.0) for k in range(1, K)}
    ofs = 0
    for s in sc:
        nx, ny = int(s["nx"]), int(s["ny"])
        n = nx * ny
        codes = rd.codes[ofs:ofs + n].reshape(ny, nx)
        vis = rd.visited[ofs:ofs + n].reshape(ny, nx).astype(bool)
        ofs += n
        if int(s["ystep"]) != 2:
            continue
        for k in range(1, K):
            m = ((codes == 1) | (codes <= -k)) & vis
            for ty in range(0, ny, 8):
                for tx in range(0, nx, 64):
                    t = m[ty:ty + 8, tx:tx + 64]
                    if not t.any():
                        continue
                    ly, lx = np.nonzero(t)
                    c1 = np.bincount((lx + 24 * ly) & 31, minlength=32)      # TileGeom<2>: skew 24 for 24x24 windows
                    fy, fx = [], []
                    for y in range(t.shape[0]):
                        row, x = t[y], 0
                        while x < row.size:
                            if row[x]:
                                fy.append(y)
                                fx.append(x)
                                x += 2 if (x + 1 < row.size and row[x + 1]) else 1
                            else:
                                x += 1
                    cp = np.bincount((np.array(fx) + 21 * np.array(fy)) & 31, minlength=32)  # TileGeomP: skew 21
                    st = stat[k]
                    st["tiles"] += 1
                    st["n"] += int(t.sum())
                    st["slots"] += len(fx)
                    st["r1"] += int(c1.max())
                    st["rp"] += int(cp.max())
                    st["p1"] += passes(int(c1.max()), 4)
                    st["pp"] += passes(int(cp.max()), 2)
    print("stage  tiles  windows/tile  ideal rows (n/32)  rows R  passes | slots/tile  rows R (pair)  passes (pair)")
    for k in range(1, K):
        st = stat[k]
        T = st["tiles"]
        print(f"{k:5d} {T:6d} {st['n'] / T:13.1f} {st['n'] / T / 32:18.2f} {st['r1'] / T:7.2f} {st['p1'] / T:7.2f} | {st['slots'] / T:10.1f} "
              f"{st['rp'] / T:14.2f} {st['pp'] / T:14.2f}")


if __name__ == "__main__":
    main()
