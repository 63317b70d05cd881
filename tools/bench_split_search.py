#!/usr/bin/env python3
"""Split search at BASELINE.json configs[4] scale: Haar BASIC, 162 336 variables x
20 000 samples (10 000 + 10 000, seed 7). Times cc_eval_presort (once per stage) and cc_eval_find_best_split (once per
tree node: Gentle AdaBoost regression stump = one call per weak learner), and checks the winner against the CPU oracle
run on a window of variables around it; the oracle's time per variable is extrapolated to the whole catalog.
usage: bench_split_search.py [HAAR|LBP] [n_samples] ; prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.bench_training_eval import samples  # noqa: E402


def main():
    import cascadeclassifier_amd as cc
    from cascadeclassifier_amd import evaluator as ev
    from oracle import oracle as orc

    kind = sys.argv[1] if len(sys.argv) > 1 else "HAAR"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    ftype = ev.HAAR if kind == "HAAR" else ev.LBP
    imgs, labels = samples(n=N // 2)
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, ev.BASIC), N, (24, 24))
    e.setImages(imgs, labels)
    F = e.getNumFeatures()
    t0 = time.perf_counter()
    e.presort()
    t_presort = time.perf_counter() - t0
    t0 = time.perf_counter()
    e.presort()
    t_presort2 = time.perf_counter() - t0

    lab = labels.astype(np.int32)
    resp = (lab * 2 - 1).astype(np.float32)
    rng = np.random.default_rng(1)
    w = rng.random(N) + 0.05
    w /= w.sum()
    tot = float(np.cumsum(w)[-1])  # sequential double sum, as calc_node_value accumulates it
    W = np.concatenate([w, [tot, 0.0]])
    nv = float(np.cumsum(resp * w)[-1] * (1.0 / tot))
    res = {}
    for name, bt, kw in (("gentle_reg", ev.BOOST_GENTLE, {"responses": resp, "node_value": nv}),
                         ("real_gini", ev.BOOST_REAL, {"class_labels": lab}),
                         ("discrete_misclass", ev.BOOST_DISCRETE, {"class_labels": lab})):
        Wk = W if bt == ev.BOOST_GENTLE else np.concatenate([w, [float(np.cumsum(w * (lab == 0))[-1]), float(np.cumsum(w * (lab == 1))[-1])]])
        e.find_best_split(Wk, boost_type=bt, **kw)  # warm-up
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            got = e.find_best_split(Wk, boost_type=bt, **kw)
        dt = (time.perf_counter() - t0) / reps
        res[name] = {"wall_ms": round(dt * 1e3, 3), "kernel_ms": round(e.last_kernel_ms(), 3), "var_idx": got["var_idx"],
                     "quality": float(got["quality"])}
        if name == "gentle_reg":
            best = got
    # oracle on a window of variables that contains the winner
    f0 = max(0, min(F - 256, best["var_idx"] - 128))
    f1 = f0 + 256
    s, t, nf = orc.set_images(imgs, want_tilted=False, want_norm=ftype == ev.HAAR)
    if ftype == ev.HAAR:
        cat = orc.haar_catalog(24, 24, ev.BASIC)
        t0 = time.perf_counter()
        vals = orc.haar_eval_batch(cat, f0, f1, s, t, nf, 24, 24)
    else:
        cat = orc.lbp_catalog(24, 24)
        t0 = time.perf_counter()
        vals = orc.lbp_eval_batch(cat, f0, f1, s, 24, 24)
    want = orc.find_best_split(vals, W, categorical=ftype == ev.LBP, responses=resp, node_value=nv)
    t_cpu = time.perf_counter() - t0
    ok = bool(want["found"]) and bool(want["var_idx"] + f0 == best["var_idx"]) and bool(want["quality"] == best["quality"]) and \
        bool(want["ord_c"] == best["ord_c"]) and bool((want["subset"] == best["subset"]).all())
    table_bytes = F * N * (6 if ftype == ev.HAAR else 1)
    out = {"workload": f"{kind} BASIC 24x24: {F} variables x {N} samples", "presort_s": round(t_presort, 3), "presort_again_s": round(t_presort2, 3),
           "resident_table_GB": round(table_bytes / 1e9, 2), "find_best_split": res,
           "table_GBps_gentle": round(table_bytes / (res["gentle_reg"]["kernel_ms"] * 1e-3) / 1e9, 1),
           "cpu_oracle_s_per_256_vars": round(t_cpu, 3), "cpu_oracle_extrapolated_s_all_vars_1_thread": round(t_cpu * F / 256, 1),
           "winner_matches_oracle": ok}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
