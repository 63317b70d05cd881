#!/usr/bin/env python3
"""Generates data/haarcascade_frontalface_synthetic.xml: a SYNTHETIC Haar cascade with the stage profile of OpenCV's
stock haarcascade_frontalface_default.xml (24x24, BASIC features, stumps, 25 stages, weak counts below = 2913), which is
not available in this container. The file is in the new cascade.xml format that
CvCascadeClassifier::save writes.

Construction (deterministic, seed 42):
  * features are drawn from the BASIC catalog of the oracle (reference order, haarfeatures.cpp:127-251);
  * each stump threshold is the median of its normalised feature value over calibration windows cut from natural-like
    1/f-noise frames at several pyramid scales;
  * leaf values +-a (a in [0.25, 1]) are signed so that a fixed synthetic "face" template collects +a from every stump;
  * each stage threshold is the median stage sum over the calibration windows that passed all previous stages
    (>= 2000 of them, otherwise over all), so each stage passes about half of what reaches it, while the template and
    near copies of it pass everything.
It is a test/benchmark fixture, not a trained detector.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from tests.util import frame_natural  # noqa: E402

STAGE_WEAK = [9, 16, 27, 32, 52, 53, 62, 72, 83, 91, 99, 115, 127, 135, 136, 137, 159, 155, 169, 196, 197, 181, 199, 211, 200]
W = H = 24


def face_template() -> np.ndarray:
    """A smooth 24x24 face-like pattern (bright face, dark eye band, bright nose bridge, dark mouth)."""
    y, x = np.mgrid[0:24, 0:24].astype(np.float64)
    img = 150 + 40 * np.exp(-(((x - 11.5) / 9) ** 2 + ((y - 11.5) / 11) ** 2))
    for cx in (6.5, 16.5):
        img -= 90 * np.exp(-(((x - cx) / 2.6) ** 2 + ((y - 8.0) / 1.8) ** 2))
    img += 35 * np.exp(-(((x - 11.5) / 1.6) ** 2 + ((y - 11.0) / 4.0) ** 2))
    img -= 70 * np.exp(-(((x - 11.5) / 4.2) ** 2 + ((y - 18.0) / 1.5) ** 2))
    img -= 25 * np.exp(-(((y - 1.0) / 2.0) ** 2))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def calibration_windows(n, rng):
    wins = []
    frames = [frame_natural(640, 480, 1000 + s) for s in range(16)]
    scales = [1.0, 1.5, 2.3, 3.4, 5.0]
    per = n // (len(frames) * len(scales)) + 1
    for f in frames:
        for s in scales:
            dw, dh = int(round(640 / s)), int(round(480 / s))
            img = orc.resize_linear_exact(f, dw, dh)
            ys = rng.integers(0, dh - H, per)
            xs = rng.integers(0, dw - W, per)
            for yy, xx in zip(ys, xs):
                wins.append(img[yy:yy + H, xx:xx + W])
    wins = np.stack(wins)[:n]
    return wins


def fmt_weight(w):
    return ("%d." % int(w)) if float(w).is_integer() else ("%.8e" % w)


def main(out_path):
    rng = np.random.default_rng(42)
    catalog = orc.haar_catalog(W, H, 0)
    area0 = catalog["r"][:, 0, 2] * catalog["r"][:, 0, 3]
    pool = np.nonzero(area0 >= 16)[0]
    n_weak = sum(STAGE_WEAK)
    chosen = rng.choice(pool, n_weak, replace=False)
    feats = catalog[chosen].copy()

    wins = calibration_windows(400000, rng)
    tmpl = face_template()
    allw = np.concatenate([wins, tmpl[None]])
    s, t, nf = orc.set_images(allw)
    ti = len(allw) - 1  # index of the template
    ok = np.nonzero(nf[:-1] > 0)[0].astype(np.int32)

    # stump thresholds: unconditional median over a 20k subset; template value decides the leaf signs
    sub = np.concatenate([ok[:20000], [ti]]).astype(np.int32)
    vals = orc.haar_eval_batch(feats, 0, n_weak, s, t, nf, W, H, sample_idx=sub)
    vt = vals[:, -1]
    thr = np.median(vals[:, :-1], axis=1).astype(np.float32)
    del vals
    a = rng.uniform(0.25, 1.0, n_weak).astype(np.float32)
    # detector rule: value < thr -> left. The template must collect +a from every stump.
    left = np.where(vt < thr, a, -a).astype(np.float32)
    right = (-left).astype(np.float32)

    # stage thresholds: median stage sum over the windows still alive (each stage passes ~half of what reaches it);
    # once fewer than 64 calibration windows are alive, 40 % of the template's (maximal) sum.
    alive = ok
    stage_thr = []
    k = 0
    for nw in STAGE_WEAK:
        if len(alive) >= 64:
            v = orc.haar_eval_batch(feats, k, k + nw, s, t, nf, W, H, sample_idx=alive)
            votes = np.where(v < thr[k:k + nw, None], left[k:k + nw, None], right[k:k + nw, None]).astype(np.float64)
            sums = votes.sum(0)
            st = np.float32(np.median(sums))
            alive = alive[sums >= st]
        else:
            st = np.float32(0.4 * a[k:k + nw].sum())
        stage_thr.append(st)
        print("stage %2d: %3d stumps, threshold %9.4f, calibration windows alive %d" % (len(stage_thr) - 1, nw, st, len(alive)),
              file=sys.stderr)
        k += nw

    L = []
    L.append('<?xml version="1.0"?>')
    L.append("<!-- SYNTHETIC cascade with the stage profile of OpenCV's haarcascade_frontalface_default.xml. -->")
    L.append("<!-- Generated by tools/make_synthetic_haar_cascade.py (seed 42). Benchmark/test fixture, not a trained detector. -->")
    L.append("<opencv_storage>")
    L.append("<cascade>")
    L.append("  <stageType>BOOST</stageType>")
    L.append("  <featureType>HAAR</featureType>")
    L.append("  <height>%d</height>" % H)
    L.append("  <width>%d</width>" % W)
    L.append("  <stageParams>")
    L.append("    <boostType>GAB</boostType>")
    L.append("    <minHitRate>9.9500000476837158e-01</minHitRate>")
    L.append("    <maxFalseAlarm>5.0000000000000000e-01</maxFalseAlarm>")
    L.append("    <weightTrimRate>9.4999999999999996e-01</weightTrimRate>")
    L.append("    <maxDepth>1</maxDepth>")
    L.append("    <maxWeakCount>%d</maxWeakCount></stageParams>" % max(STAGE_WEAK))
    L.append("  <featureParams>")
    L.append("    <maxCatCount>0</maxCatCount>")
    L.append("    <featSize>1</featSize>")
    L.append("    <mode>BASIC</mode></featureParams>")
    L.append("  <stageNum>%d</stageNum>" % len(STAGE_WEAK))
    L.append("  <stages>")
    k = 0
    for si, nw in enumerate(STAGE_WEAK):
        L.append("    <!-- stage %d -->" % si)
        L.append("    <_>")
        L.append("      <maxWeakCount>%d</maxWeakCount>" % nw)
        L.append("      <stageThreshold>%.8e</stageThreshold>" % stage_thr[si])
        L.append("      <weakClassifiers>")
        for i in range(nw):
            L.append("        <_>")
            L.append("          <internalNodes>")
            L.append("            0 -1 %d %.8e</internalNodes>" % (k, thr[k]))
            L.append("          <leafValues>")
            L.append("            %.8e %.8e</leafValues></_>" % (left[k], right[k]))
            k += 1
        L.append("      </weakClassifiers></_>")
    L.append("  </stages>")
    L.append("  <features>")
    for f in feats:
        L.append("    <_>")
        L.append("      <rects>")
        for j in range(3):
            if f["wt"][j] == 0:
                continue
            r = f["r"][j]
            L.append("        <_>")
            L.append("          %d %d %d %d %s</_>" % (r[0], r[1], r[2], r[3], fmt_weight(f["wt"][j])))
        L[-1] += "</rects>"
        L.append("      <tilted>0</tilted></_>")
    L.append("  </features>")
    L.append("</cascade>")
    L.append("</opencv_storage>")
    with open(out_path, "w") as fh:
        fh.write("\n".join(L) + "\n")
    np.save(os.path.join(ROOT, "data", "face_template_24x24.npy"), tmpl)
    print("wrote", out_path, file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml"))
