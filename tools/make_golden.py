#!/usr/bin/env python3
"""Generates tests/golden/hotpath_golden.npz: input/expected-output vectors for the hot path, produced by the CPU oracle
(oracle/cc_oracle.cpp) in this container. The reference itself cannot run here (it needs OpenCV, absent), so these
vectors pin the ORACLE'S restatement: the training-side values are anchored to the reference's own KATs
(tests/test_oracle_kats.py); the detection-side values are 'parity unpinned' against OpenCV. Inputs are regenerated
from seeds by tests/util.py; only outputs (and small inputs) are stored."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from tests.util import frame_natural, read_vec, upscale  # noqa: E402


def golden_frame():
    tm = np.load(os.path.join(ROOT, "data", "face_template_24x24.npy"))
    img = frame_natural(320, 240, 77)
    for k, (x, y) in zip((1.0, 1.9, 3.1), ((20, 30), (150, 40), (210, 120))):
        s = int(24 * k)
        img[y:y + s, x:x + s] = upscale(tm, s)
    return img


def main():
    out = {}
    img = golden_frame()
    out["frame_sha256"] = np.frombuffer(hashlib.sha256(img.tobytes()).digest(), np.uint8)
    for name, xml in (("haar", "haarcascade_frontalface_synthetic.xml"), ("lbp", "lbpcascade_frontalface.xml")):
        c = orc.load_cascade_xml(os.path.join(ROOT, "data", xml))
        r = orc.detect_raw(c, img, 1.1, nthreads=4, full=True)
        out[f"{name}_candidates"] = r.candidates
        out[f"{name}_codes_sha256"] = np.frombuffer(hashlib.sha256(r.codes.tobytes()).digest(), np.uint8)
        out[f"{name}_sums_sha256"] = np.frombuffer(hashlib.sha256(r.sums.tobytes()).digest(), np.uint8)
        out[f"{name}_visited_sha256"] = np.frombuffer(hashlib.sha256(r.visited.tobytes()).digest(), np.uint8)
        out[f"{name}_code_hist"] = np.bincount(r.codes + 32, minlength=40)
        out[f"{name}_rects_1p1_3"] = orc.detect_multiscale(c, img, 1.1, 3)
        out[f"{name}_rects_4_1"] = orc.detect_multiscale(c, img, 4.0, 1)
        out[f"{name}_rects_4_50"] = orc.detect_multiscale(c, img, 4.0, 50)  # tools/detection/Cpp/main.cpp:45
    # training side: barcode.vec samples, LBP 75x32 (first 4096 features) and Haar BASIC 75x32 (4096 features)
    samples = read_vec(os.path.join(ROOT, "tests", "golden", "barcode.vec"))[:8]
    s, _, _ = orc.set_images(samples, want_norm=False)
    out["lbp75x32_first4096"] = orc.lbp_eval_batch(orc.lbp_catalog(75, 32), 0, 4096, s, 75, 32).astype(np.uint8)
    s, t, nf = orc.set_images(samples)
    feats = orc.haar_catalog(75, 32, 0)
    out["haar75x32_normfactor"] = nf
    out["haar75x32_feat_1000000_1004096"] = orc.haar_eval_batch(feats, 1000000, 1004096, s, t, nf, 75, 32)
    out["haar75x32_sum_sample3"] = s[3]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hotpath_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
