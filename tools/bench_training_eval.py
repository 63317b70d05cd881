#!/usr/bin/env python3
"""BASELINE.json configs[4]: CvCascadeBoost single-stage feature evaluation — Haar BASIC (162 336 features) over
10 000 positives + 10 000 negatives of 24x24 (seed 7). Measures (i) batched setImage,
(ii) the full feature x sample matrix (12.99 GB of float32, written to HBM in row blocks) and compares a slice with the
CPU oracle, which is also timed (the reference publishes ~35 M evals/s for its precalculation, res/README.md:91,97).
Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def samples(seed=7, n=10000):
    rng = np.random.default_rng(seed)
    tmpl = rng.integers(0, 256, (24, 24)).astype(np.float64)
    pos = np.clip(np.rint(tmpl + rng.normal(0, 15, (n, 24, 24))), 0, 255).astype(np.uint8)
    neg = rng.integers(0, 256, (n, 24, 24), dtype=np.uint8)
    return np.concatenate([pos, neg]), np.concatenate([np.ones(n, np.uint8), np.zeros(n, np.uint8)])


def main():
    import torch

    import cascadeclassifier_amd as cc
    from cascadeclassifier_amd import evaluator as ev
    from oracle import oracle as orc

    mode = {"BASIC": ev.BASIC, "CORE": ev.CORE, "ALL": ev.ALL}[sys.argv[1] if len(sys.argv) > 1 else "BASIC"]
    imgs, labels = samples()
    N = len(imgs)
    e = cc.CvFeatureEvaluator.create(ev.HAAR)
    e.init(cc.CvFeatureParams(ev.HAAR, mode), N, (24, 24))
    nfeat = e.getNumFeatures()
    e.setImages(imgs[:64], labels[:64])  # warm-up
    t0 = time.perf_counter()
    e.setImages(imgs, labels)
    t_set = time.perf_counter() - t0

    block = 32768
    # does the rate of the bulk evaluator depend on the pitch of the output rows? The trainer's layout
    # is out[feature][sample] (one row per feature, o_cvcascadeboosttraindata.cpp:582-596): every wavefront stores a 256-byte
    # piece `pitch` floats away from the previous feature's. 20 000 floats = 80 000 B = 2^7 x 625; padded pitches move the
    # rows onto other channel / bank phases.
    pitches = [int(v) for v in os.environ.get("CCAMD_BENCH_PITCHES", "20000,20032,20096,20480").split(",")]
    sweep = {}
    out = torch.empty((block, max(pitches + [N])), dtype=torch.float32, device="cuda")
    for pitch in pitches:
        e.calc_batch_device(0, block, out.data_ptr(), n_samples=N, pitch=pitch)  # warm-up
        torch.cuda.synchronize()
        ms = 0.0
        for f0 in range(0, nfeat, block):
            f1 = min(f0 + block, nfeat)
            e.calc_batch_device(f0, f1, out.data_ptr(), n_samples=N, pitch=pitch)
            ms += e.last_kernel_ms()
        sweep[str(pitch)] = {"kernel_ms": round(ms, 3), "write_TBps": round(nfeat * N * 4 / (ms * 1e-3) / 1e12, 3)}
    out = torch.empty((block, N), dtype=torch.float32, device="cuda")
    e.calc_batch_device(0, block, out.data_ptr(), n_samples=N)  # warm-up
    torch.cuda.synchronize()
    kernel_ms = 0.0
    t0 = time.perf_counter()
    for f0 in range(0, nfeat, block):
        f1 = min(f0 + block, nfeat)
        e.calc_batch_device(f0, f1, out.data_ptr(), n_samples=N)
        kernel_ms += e.last_kernel_ms()
    torch.cuda.synchronize()
    t_eval = time.perf_counter() - t0

    # parity of one block against the oracle + CPU timing on the same block
    f0, f1 = 100000, 100000 + 2048
    e.calc_batch_device(f0, f1, out.data_ptr(), n_samples=N)
    torch.cuda.synchronize()
    got = out[: f1 - f0].cpu().numpy()
    feats = orc.haar_catalog(24, 24, mode)
    s, t, nf = orc.set_images(imgs, want_tilted=(mode == ev.ALL))
    t0 = time.perf_counter()
    want = orc.haar_eval_batch(feats, f0, f1, s, t, nf, 24, 24)
    t_cpu = time.perf_counter() - t0
    same = bool((got.view(np.uint32) == want.view(np.uint32)).all())
    evals = nfeat * N
    print(json.dumps({
        "workload": f"Haar {['BASIC', 'CORE', 'ALL'][mode]} {nfeat} features x {N} samples of 24x24 (seed 7)",
        "set_images_ms": round(t_set * 1e3, 3),
        "eval_matrix_wall_s": round(t_eval, 4),
        "eval_matrix_kernel_ms": round(kernel_ms, 3),
        "gevals_per_s_kernel": round(evals / (kernel_ms * 1e-3) / 1e9, 2),
        "hbm_write_GBps_kernel": round(evals * 4 / (kernel_ms * 1e-3) / 1e9, 1),
        "frac_of_8TBps": round(evals * 4 / (kernel_ms * 1e-3) / 8e12, 4),
        "slice_bit_identical_to_oracle": same,
        "row_pitch_sweep": sweep,
        "cpu_oracle_mevals_per_s_1thread": round((f1 - f0) * N / t_cpu / 1e6, 1),
    }))


if __name__ == "__main__":
    main()
