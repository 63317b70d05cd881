#!/usr/bin/env python3
"""Single-image latency of detectMultiScale through the C ABI (host image in, rectangles out), the detection tool's call
shape: config 1 of SURVEY 8d (640x480, scaleFactor 4, minNeighbors 50) and one Full-HD frame at scaleFactor 1.1."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import cascadeclassifier_amd as cc
    from tests.util import frame_natural
    out = {}
    for spec in (0, 7):
        clf = cc.CascadeClassifier(os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"))
        if spec:
            clf.specialize(spec)
        for name, (w, h, sf, mn) in {"640x480_sf4_mn50": (640, 480, 4.0, 50), "640x480_sf1.1": (640, 480, 1.1, 3),
                                     "1920x1080_sf1.1": (1920, 1080, 1.1, 3)}.items():
            img = frame_natural(w, h, 5)
            for _ in range(5):
                clf.detectMultiScale(img, sf, mn)
            ts = []
            for _ in range(50):
                t0 = time.perf_counter()
                r = clf.detectMultiScale(img, sf, mn)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            out[f"{name}_spec{spec}"] = {"median_ms": round(ts[len(ts) // 2] * 1e3, 3), "p90_ms": round(ts[int(len(ts) * 0.9)] * 1e3, 3), "rects": len(r)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
