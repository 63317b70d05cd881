#!/usr/bin/env python3
"""Batched negative mining (cc_negminer_run) on one 1920x1080 background image with the first K stages of
the synthetic Haar cascade (the trainer's situation while stage K is being filled): every window of the reader's
sqrt(2) / half-window-step stream through the trained stages. Prints one JSON line; run under rocprofv3 for kernel stats.
usage: bench_negmine.py [K=10] [reps=5]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import cascadeclassifier_amd as cc
    from cascadeclassifier_amd import evaluator as ev
    from tests.util import frame_natural
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    xml = os.path.join(tempfile.mkdtemp(), "trunc.xml")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "truncate_cascade.py"),
                           os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"), str(K), xml])
    c = cc.CascadeClassifier(xml)
    m = ev.NegativeMiner(c)
    img = frame_natural(1920, 1080, 5)
    plan = m.plan(1920, 1080)
    flags, pix, idx = m.run(img, max_keep=256)  # warm-up
    t0 = time.perf_counter()
    for _ in range(reps):
        flags, pix, idx = m.run(img, max_keep=256)
    dt = (time.perf_counter() - t0) / reps
    out = {"workload": f"negative mining, 1920x1080 background, {K} trained stages, {plan['n_windows']} stream windows in "
                       f"{len(plan['levels'])} ladder levels", "wall_ms_per_image": round(dt * 1e3, 3),
           "mwindows_per_s": round(plan["n_windows"] / dt / 1e6, 2), "accepted": int(flags.sum())}
    # cc_negminer_run_batch: B images of one size per call (consecutive images of a background set)
    batches = {}
    for (w, h) in ((1920, 1080), (640, 480)):
        per = m.plan(w, h)["n_windows"]
        for B in (1, 8, 32):
            imgs = [frame_natural(w, h, 100 + k) for k in range(B)]
            fb = m.run_batch(imgs, max_keep=256)[0]  # warm-up (sizes the workspace)
            t0 = time.perf_counter()
            for _ in range(reps):
                fb = m.run_batch(imgs, max_keep=256)[0]
            dtb = (time.perf_counter() - t0) / reps
            one = m.run(imgs[-1], max_keep=256)[0]
            batches[f"{w}x{h} x {B}"] = {"wall_ms_per_call": round(dtb * 1e3, 3), "wall_ms_per_image": round(dtb / B * 1e3, 4),
                                          "mwindows_per_s": round(per * B / dtb / 1e6, 1), "last_image_equals_single_call": bool((fb[-1] == one).all())}
    out["run_batch"] = batches
    print(json.dumps(out))


if __name__ == "__main__":
    main()
