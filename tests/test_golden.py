"""Committed golden vectors (tests/golden/hotpath_golden.npz, made by tools/make_golden.py with the oracle).
CPU: the oracle still reproduces them (guards the oracle against drift). GPU: the HIP path reproduces them through the
C ABI without the oracle in the loop."""
import hashlib
import os

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import evaluator as ev
from oracle import oracle as orc
from tests.util import read_vec
from tools.make_golden import golden_frame

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_golden.npz"))
XML = {"haar": "haarcascade_frontalface_synthetic.xml", "lbp": "lbpcascade_frontalface.xml"}


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_golden_frame_is_reproducible():
    assert (_sha(golden_frame()) == G["frame_sha256"]).all()


@pytest.mark.parametrize("name", ["haar", "lbp"])
def test_oracle_reproduces_golden(repo_root, name):
    c = orc.load_cascade_xml(os.path.join(repo_root, "data", XML[name]))
    img = golden_frame()
    r = orc.detect_raw(c, img, 1.1, nthreads=2, full=True)
    assert (r.candidates == G[f"{name}_candidates"]).all()
    assert (_sha(r.codes) == G[f"{name}_codes_sha256"]).all() and (_sha(r.sums) == G[f"{name}_sums_sha256"]).all()
    assert (orc.detect_multiscale(c, img, 1.1, 3) == G[f"{name}_rects_1p1_3"]).all()
    assert len(G["haar_rects_1p1_3"]) >= 3  # the pasted faces are found


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["haar", "lbp"])
def test_gpu_reproduces_golden_detection(repo_root, name):
    p = cc.CascadeClassifier(os.path.join(repo_root, "data", XML[name]))
    img = golden_frame()
    codes, sums, vis = p.debug_windows(img, 1.1)
    assert (_sha(codes) == G[f"{name}_codes_sha256"]).all()
    assert (_sha(sums) == G[f"{name}_sums_sha256"]).all()  # stage sums bit-identical
    assert (_sha(vis) == G[f"{name}_visited_sha256"]).all()
    assert (np.bincount(codes + 32, minlength=40) == G[f"{name}_code_hist"]).all()
    assert (p.detect_raw(img, 1.1) == G[f"{name}_candidates"]).all()
    for key, (sf, mn) in (("rects_1p1_3", (1.1, 3)), ("rects_4_1", (4.0, 1)), ("rects_4_50", (4.0, 50))):
        got = p.detectMultiScale(img, sf, mn)
        want = G[f"{name}_{key}"]
        assert got.shape == want.shape and (got == want).all()


@pytest.mark.gpu
def test_gpu_reproduces_golden_feature_values(repo_root):
    samples = read_vec(os.path.join(repo_root, "tests", "golden", "barcode.vec"))[:8]
    e = cc.CvFeatureEvaluator.create(ev.LBP)
    e.init(cc.CvFeatureParams(ev.LBP), 8, (75, 32))
    e.setImages(samples)
    assert (e.calc_batch(0, 4096).astype(np.uint8) == G["lbp75x32_first4096"]).all()
    h = cc.CvFeatureEvaluator.create(ev.HAAR)
    h.init(cc.CvFeatureParams(ev.HAAR, ev.BASIC), 8, (75, 32))
    h.setImages(samples)
    got = h.calc_batch(1000000, 1004096)
    assert (got.view(np.uint32) == G["haar75x32_feat_1000000_1004096"].view(np.uint32)).all()
    s3, _, nf3 = h.get_sample(3)
    assert (s3 == G["haar75x32_sum_sample3"]).all() and np.float32(nf3) == G["haar75x32_normfactor"][3]
