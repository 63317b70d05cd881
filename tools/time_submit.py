#!/usr/bin/env python3
"""Host-side durations of cc_detect_batch_submit / _collect in the bench's pipelined loop (64 Full-HD frames per batch)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cascadeclassifier_amd as cc
from bench import make_frames
B, H, W = 64, 1080, 1920
fr = make_frames(B, W, H, 0)
frames = torch.from_numpy(fr).cuda()
clf = cc.CascadeClassifier(os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"), max_batch=B)
clf.specialize(7)
clf.detect_batch(None, 1.1, 3, device_ptr=frames.data_ptr(), shape=(B, H, W))
HOST = len(sys.argv) > 1 and sys.argv[1] == "host"  # frames in pageable host memory (pinned staging inside submit)


def submit():
    if HOST:
        return clf.detect_batch_submit(fr, 1.1, 3)
    return clf.detect_batch_submit(None, 1.1, 3, device_ptr=frames.data_ptr(), shape=(B, H, W))


for _ in range(3):  # warm-up: buffers sized, pipeline full
    w = submit()
    clf.detect_batch_collect(w)
prev = None
t00 = time.perf_counter()
for i in range(8):
    t0 = time.perf_counter()
    t = submit()
    t1 = time.perf_counter()
    if prev is not None:
        clf.detect_batch_collect(prev)
    t2 = time.perf_counter()
    prev = t
    print("iter %d: submit starts %.2f ms, takes %.2f ms; collect takes %.2f ms" % (i, (t0 - t00) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3))
clf.detect_batch_collect(prev)
print("%s frames: %.3f ms per step over 8 pipelined steps" % ("host" if HOST else "resident", (time.perf_counter() - t00) / 8 * 1e3))
