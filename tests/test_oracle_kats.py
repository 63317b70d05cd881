"""Pins the CPU oracle to every known-answer value the reference holds for the hot path.

Sources (paths relative to /root/reference; values are copied as DATA, not code):
  * traincascade/test/test_features.cpp:462-560  Feature::calc KATs  -3200 / 0 / -3600 / 32 (tilted)
  * traincascade/test/test_features.cpp:252-362  constant-image invariants (Haar == 0, LBP == 255), step-edge images
  * traincascade/res/README.md:41,91             catalog sizes 152 625 (LBP 75x32) and 2 790 554 (Haar BASIC 75x32)
  * traincascade/res/barcode.vec                 fixture (sha256 f76315c9...76db), copied to tests/golden/
The detection-side functions of the oracle (resize, detectMultiScale, groupRectangles) restate OpenCV 4.6.0, which is
absent here: PARITY UNPINNED for those (self-consistency and hand-computed cases only, tests/test_oracle_detect.py).
"""
import hashlib
import os

import numpy as np

from oracle import oracle as orc


def _flat_integral(img):
    return orc.integral(img)["sum"].reshape(-1)


def test_calc_kat_upright_two_rect_step_image():  # test_features.cpp:462-483
    img = np.zeros((8, 8), np.uint8)
    img[:, 4:8] = 100
    f = orc.make_haar_feature(False, [(0, 0, 4, 8, +1.0), (4, 0, 4, 8, -1.0)])
    assert orc.haar_feature_calc(f, _flat_integral(img), 9) == -3200.0


def test_calc_kat_uniform_is_zero():  # test_features.cpp:485-502
    img = np.full((8, 8), 42, np.uint8)
    f = orc.make_haar_feature(False, [(0, 0, 4, 8, +1.0), (4, 0, 4, 8, -1.0)])
    assert orc.haar_feature_calc(f, _flat_integral(img), 9) == 0.0


def test_calc_kat_three_rect():  # test_features.cpp:504-529
    img = np.zeros((3, 9), np.uint8)
    img[:, 3:6] = 200
    f = orc.make_haar_feature(False, [(0, 0, 9, 3, +1.0), (3, 0, 3, 3, -3.0)])
    assert orc.haar_feature_calc(f, _flat_integral(img), 10) == -3600.0


def test_calc_kat_tilted():  # test_features.cpp:531-560
    img = np.ones((16, 16), np.uint8)
    tilted = orc.integral(img, tilted=True)["tilted"].reshape(-1)
    f = orc.make_haar_feature(True, [(8, 2, 4, 4, +1.0)])
    assert orc.haar_feature_calc(f, tilted, 17) == 32.0


def test_catalog_sizes():  # res/README.md:41,91 + SURVEY.md §4 (24x24 re-derivations)
    assert orc.lbp_catalog(75, 32).shape[0] == 152625
    assert orc.haar_catalog_size(75, 32, 0) == 2790554
    assert orc.haar_catalog_size(24, 24, 0) == 162336
    assert orc.haar_catalog_size(24, 24, 1) == 210400
    assert orc.haar_catalog_size(24, 24, 2) == 261600
    assert orc.lbp_catalog(24, 24).shape[0] == 8464
    assert orc.haar_catalog_size(24, 24, 2) > orc.haar_catalog_size(24, 24, 0)  # test_features.cpp:150-223


def test_haar_constant_image_all_zero():  # test_features.cpp:252-275, 300-317
    for mode, val in ((0, 128), (2, 64)):
        feats = orc.haar_catalog(24, 24, mode)
        s, t, nf = orc.set_images(np.full((1, 24, 24), val, np.uint8), want_tilted=(mode == 2))
        assert nf[0] == 0.0
        v = orc.haar_eval_batch(feats, 0, len(feats), s, t, nf, 24, 24)
        assert not v.any()


def test_haar_step_edge_nonzero():  # test_features.cpp:277-298
    img = np.zeros((1, 24, 24), np.uint8)
    img[0, :, 12:] = 255
    feats = orc.haar_catalog(24, 24, 0)
    s, t, nf = orc.set_images(img)
    assert nf[0] > 0
    assert orc.haar_eval_batch(feats, 0, len(feats), s, t, nf, 24, 24).any()


def test_lbp_constant_image_all_255():  # test_features.cpp:319-340
    rects = orc.lbp_catalog(24, 24)
    s, _, _ = orc.set_images(np.full((1, 24, 24), 50, np.uint8), want_norm=False)
    v = orc.lbp_eval_batch(rects, 0, len(rects), s, 24, 24)
    assert (v == 255.0).all()


def test_lbp_step_edge_and_sample_isolation():  # test_features.cpp:342-392
    imgs = np.zeros((2, 24, 24), np.uint8)
    imgs[0] = 80
    imgs[1, 12:, :] = 200
    rects = orc.lbp_catalog(24, 24)
    s, _, _ = orc.set_images(imgs, want_norm=False)
    v = orc.lbp_eval_batch(rects, 0, len(rects), s, 24, 24)
    assert (v[:, 0] == 255.0).all()
    assert (v[:, 1] < 255.0).any()


def test_integral_matches_numpy_cumsum():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    r = orc.integral(img, sqsum_f64=True, sqsum_i32=True, tilted=True)
    ref = np.zeros((38, 54), np.int64)
    ref[1:, 1:] = img.astype(np.int64).cumsum(0).cumsum(1)
    assert (r["sum"] == ref).all()
    refq = np.zeros((38, 54), np.int64)
    refq[1:, 1:] = (img.astype(np.int64) ** 2).cumsum(0).cumsum(1)
    assert (r["sqsum_f64"] == refq).all()
    assert (r["sqsum_i32"].view(np.uint32) == (refq & 0xFFFFFFFF)).all()
    # tilted by the definition, brute force
    t = np.zeros((38, 54), np.int64)
    for Y in range(38):
        for X in range(54):
            acc = 0
            for y in range(Y):
                half = Y - y - 1
                x0, x1 = max(X - 1 - half, 0), min(X - 1 + half, 52)
                if x1 >= x0:
                    acc += int(img[y, x0:x1 + 1].sum())
            t[Y, X] = acc
    assert (r["tilted"] == t).all()


def test_sqsum_i32_wraps_like_uint32():
    img = np.full((300, 300), 255, np.uint8)  # 300*300*65025 = 5.85e9 > 2^32
    r = orc.integral(img, sqsum_i32=True)
    assert int(r["sqsum_i32"].view(np.uint32)[300, 300]) == (300 * 300 * 65025) % (1 << 32)


def test_barcode_vec_fixture(repo_root):
    path = os.path.join(repo_root, "tests", "golden", "barcode.vec")
    raw = open(path, "rb").read()
    assert hashlib.sha256(raw).hexdigest() == "f76315c9351c82aeaabd66b066c203f358a11d1e69c1cc5906a0373320fe76db"
    from tests.util import read_vec
    samples = read_vec(path)
    assert samples.shape == (100, 32, 75)
    # Haar BASIC / LBP catalogs over real samples stay finite and LBP codes are 0..255
    rects = orc.lbp_catalog(75, 32)
    s, _, _ = orc.set_images(samples[:4], want_norm=False)
    v = orc.lbp_eval_batch(rects, 0, 2000, s, 75, 32)
    assert v.min() >= 0 and v.max() <= 255 and (v == np.floor(v)).all()
