#!/usr/bin/env python3
"""One-off stress run for the kernel generator (not part of the suite): random upright stump cascades -- random stage
counts and stage sizes from 1 stump up, random leaves -- specialised as far as the budget allows and compared with the CPU
oracle window by window (result codes, exit stages, stage sums, visited flags, rectangles) on a natural-like and a
uniform-noise frame; round 4 adds the kernel forms of that round (one module for both steps / one per step at other tile heights,
the list queue from stage 1) and random LBP stump cascades of several window sizes (16-bit tiles of 20 / 16 / 8 window rows,
list queue on / off). The shapes the generator treats differently all occur: stages shorter than the four parts of a stage,
stages whose sums are exact (fixed-point votes, stumps re-ordered to share corners) and not (float accumulation in the
cascade's order), rectangles whose sums exceed 16 bits. Usage on a GPU box: python tests/analysis/stress_specialised.py [n_cascades] [first_seed]
Last run: profiles/r04_stress_specialised.txt."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cascadeclassifier_amd as cc  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests import cascade_factory as cf  # noqa: E402
from tests.util import frame_natural, frame_uniform  # noqa: E402


def main():
    n_cascades = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    img, img2 = frame_natural(416, 300, 7), frame_uniform(200, 150, 8)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 270, 9) for x in range(0, 390, 11)])
    t0 = time.time()
    windows = 0
    for seed in range(seed0, seed0 + n_cascades):
        rng = np.random.default_rng(1000 + seed)
        sizes = tuple(int(v) for v in rng.choice([1, 2, 3, 5, 8, 13, 21, 34, 47], size=int(rng.integers(2, 7))))
        xml = cf.tilted_stump_cascade(cal, seed=seed, stage_sizes=sizes, tilted=False, min_area=int(rng.choice([16, 100, 258])))
        path = f"/tmp/stress_{seed}.xml"
        open(path, "w").write(xml)
        o = orc.load_cascade_xml(path)
        for env in ({}, {"CCAMD_SPEC_TILE16": "1"}, {"CCAMD_SPEC_PAIR16": "1"}, {"CCAMD_SPEC_ONE_MODULE": "1"},
                    {"CCAMD_SPEC_TILE_Y1": "8", "CCAMD_SPEC_TILE_Y2": "12"}, {"CCAMD_DENSE_FROM": "1"}):
            os.environ.update(env)
            p = cc.CascadeClassifier(path)
            k = p.specialize(len(sizes))
            for im, sf in ((img, 1.1), (img2, 1.3)):
                ref = orc.detect_raw(o, im, sf, nthreads=8, full=True)
                codes, sums, vis = p.debug_windows(im, sf)
                assert (codes == ref.codes).all() and (sums == ref.sums).all() and (vis == ref.visited).all(), (seed, sizes, env)
                a, b = p.detectMultiScale(im, sf, 2), orc.detect_multiscale(o, im, sf, 2, nthreads=8)
                assert a.shape == b.shape and (a == b).all(), (seed, sizes, env)
                windows += len(codes)
            for kk in env:
                del os.environ[kk]
        print(f"cascade {seed}: stages {sizes}, {k} specialised: identical (32-bit, 16-bit and pair tiles; one module, other tile heights, list queue)", flush=True)
        os.remove(path)
    for seed in range(seed0, seed0 + n_cascades):
        rng = np.random.default_rng(5000 + seed)
        W, H = [(24, 24), (20, 20), (18, 30), (32, 16)][seed % 4]
        sizes = tuple(int(v) for v in rng.choice([1, 2, 3, 4, 6, 9, 14], size=int(rng.integers(2, 8))))
        path = f"/tmp/stress_lbp_{seed}.xml"
        open(path, "w").write(cf.lbp_stump_cascade(W, H, seed=100 + seed, stage_sizes=sizes))
        o = orc.load_cascade_xml(path)
        for env in ({}, {"CCAMD_SPEC_TILE_Y": "16"}, {"CCAMD_SPEC_TILE_Y": "8"}, {"CCAMD_DENSE_FROM": "99"}, {"CCAMD_DENSE_FROM": "1"}, {"CCAMD_GROUP_STUMPS": "1"}):
            os.environ.update(env)
            p = cc.CascadeClassifier(path)
            k = p.specialize(len(sizes))
            for im, sf in ((img, 1.1), (img2, 1.3)):
                ref = orc.detect_raw(o, im, sf, nthreads=8, full=True)
                codes, sums, vis = p.debug_windows(im, sf)
                assert (codes == ref.codes).all() and (sums == ref.sums).all() and (vis == ref.visited).all(), ("lbp", seed, sizes, env)
                a, b = p.detectMultiScale(im, sf, 2), orc.detect_multiscale(o, im, sf, 2, nthreads=8)
                assert a.shape == b.shape and (a == b).all(), ("lbp", seed, sizes, env)
                windows += len(codes)
            for kk in env:
                del os.environ[kk]
        print(f"LBP cascade {seed}: window {W}x{H}, stages {sizes}, {k} specialised: identical (tiles of 20 / 16 / 8 rows, table and list queues, groups of one stage)", flush=True)
        os.remove(path)
    print(f"{n_cascades} random Haar + {n_cascades} random LBP cascades x 6 kernel forms x 2 frames: {windows} windows, every code / exit stage / stage sum / visited flag / "
          f"rectangle identical to the CPU oracle ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
