#!/usr/bin/env python3
"""BASELINE.json configs[4] end to end: one boosted stage of Gentle AdaBoost stumps (CvCascadeBoost::train with
maxDepth 1) on 10 000 + 10 000 samples of 24x24 with the full Haar BASIC catalog (162 336 variables), every
data-parallel step on the device: batched setImage, presort of all variables (once), then per weak learner the node
split search and one feature row for the sample directions. The boosting bookkeeping between those calls (node values,
w *= exp(-y f), renormalisation; boost.cpp:378-398, o_cvboostree.cpp:657-732) is the reference's serial host code,
restated here in numpy with sequential double sums. Prints one JSON line.
usage: bench_boost_stage.py [rounds=16] [n_samples=20000] [easy|hard] [HAAR|LBP]
  easy = SURVEY config 5 as specified (template + N(0,15^2) vs uniform noise: one stump separates it);
  hard = positives template + N(0,40^2), negatives a half-and-half blend with a second template + N(0,40^2)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_training_eval import samples  # noqa: E402


def seq_sum(a):
    """Left-to-right double sum (what the reference's loops compute); np.cumsum accumulates sequentially."""
    return float(np.cumsum(a, dtype=np.float64)[-1]) if len(a) else 0.0


def main():
    import cascadeclassifier_amd as cc
    from cascadeclassifier_amd import evaluator as ev
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    imgs, labels = samples(n=N // 2)
    hard = len(sys.argv) > 3 and sys.argv[3] == "hard"
    lbp = len(sys.argv) > 4 and sys.argv[4] == "LBP"  # the same stage over the 8 464 categorical LBP variables
    ftype = ev.LBP if lbp else ev.HAAR
    if hard:
        rng = np.random.default_rng(7)
        t1 = rng.integers(0, 256, (24, 24)).astype(np.float64)
        t2 = rng.integers(0, 256, (24, 24)).astype(np.float64)
        pos = np.clip(np.rint(t1 + rng.normal(0, 40, (N // 2, 24, 24))), 0, 255).astype(np.uint8)
        neg = np.clip(np.rint(0.5 * t1 + 0.5 * t2 + rng.normal(0, 40, (N - N // 2, 24, 24))), 0, 255).astype(np.uint8)
        imgs = np.concatenate([pos, neg])
    y = labels.astype(np.float64) * 2 - 1
    resp = y.astype(np.float32)
    # process start-up (HIP context, code-object load) is not stage work: a throw-away evaluator takes it first
    t0 = time.perf_counter()
    w0 = cc.CvFeatureEvaluator.create(ftype)
    w0.init(cc.CvFeatureParams(ftype, ev.BASIC), 64, (24, 24))
    w0.setImages(imgs[:64], labels[:64])
    del w0
    t_startup = time.perf_counter() - t0
    t0 = time.perf_counter()
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, ev.BASIC), N, (24, 24))
    e.setImages(imgs, labels)
    t_set = time.perf_counter() - t0
    t0 = time.perf_counter()
    e.presort()
    t_presort = time.perf_counter() - t0
    w = np.full(N, 1.0 / N)
    F = np.zeros(N)
    t_split = t_row = t_host = 0.0
    kernel_ms = []
    chosen = []
    errs = []
    for r in range(rounds):
        t0 = time.perf_counter()
        tot = seq_sum(w)
        nv = seq_sum(y * w) * (1.0 / tot)
        W = np.concatenate([w, [tot, 0.0]])
        t_host += time.perf_counter() - t0
        t0 = time.perf_counter()
        sp = e.find_best_split(W, responses=resp, node_value=nv)
        t_split += time.perf_counter() - t0
        kernel_ms.append(e.last_kernel_ms())
        assert sp["found"]
        t0 = time.perf_counter()
        v = e.calc_batch(sp["var_idx"], sp["var_idx"] + 1)[0]
        t_row += time.perf_counter() - t0
        t0 = time.perf_counter()
        if lbp:  # CV_DTREE_CAT_DIR: a sample goes left when its category's bit is set in the split's subset
            code = v.astype(np.int64)
            left = ((np.ascontiguousarray(sp["subset"], dtype=np.int32).view(np.uint32)[code >> 5] >> (code & 31).astype(np.uint32)) & 1).astype(bool)
        else:
            left = v <= sp["ord_c"]
        f = np.empty(N)
        for side in (left, ~left):
            sw = seq_sum(w[side])
            f[side] = seq_sum((y * w)[side]) * (1.0 / sw) if sw > 0 else 0.0
        F += f
        w = w * np.exp(-y * f)
        w = w * (1.0 / seq_sum(w))
        t_host += time.perf_counter() - t0
        chosen.append(int(sp["var_idx"]))
        errs.append(float(np.mean(np.sign(F) != y)))
    out = {"data": "hard" if hard else "easy (SURVEY config 5)", "workload": f"Gentle AdaBoost, {rounds} stumps, {'LBP' if lbp else 'Haar BASIC'} 24x24 ({e.getNumFeatures()} variables) x {N} samples (seed 7)",
           "process_startup_s": round(t_startup, 3), "set_images_s": round(t_set, 3), "presort_s": round(t_presort, 3),
           "per_weak_learner_ms": {"split_search_wall": round(t_split / rounds * 1e3, 3), "split_search_kernel": round(float(np.mean(kernel_ms)), 3),
                                   "feature_row": round(t_row / rounds * 1e3, 3), "host_bookkeeping_numpy": round(t_host / rounds * 1e3, 3)},
           "stage_total_s": round(t_set + t_presort + t_split + t_row + t_host, 3),
           "chosen_variables": chosen, "training_error_after_each_round": [round(x, 4) for x in errs]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
