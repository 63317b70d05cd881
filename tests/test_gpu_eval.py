"""GPU parity tests of the training-side evaluator (CvHaarEvaluator / CvLBPEvaluator replacement) through the C ABI.
The first block re-states the reference's own test cases (traincascade/test/test_features.cpp) against the HIP path;
the rest compares bulk evaluation with the oracle. LBP codes, integrals and Feature::calc are bit-exact; Haar
operator() values (one float division by the norm factor) are compared exactly as well (tolerance 0)."""
import os

import ctypes as C

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import _lib as L
from cascadeclassifier_amd import evaluator as ev
from oracle import oracle as orc
from tests.util import read_vec

pytestmark = pytest.mark.gpu


def _mk(ftype, mode, n, win=(24, 24)):
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, mode), n, win)
    return e


# ---- the reference's own cases (test_features.cpp) ------------------------------------------------------------
def test_factory_and_getters():  # test_features.cpp:15-70, 150-223
    assert cc.CvFeatureEvaluator.create(99) is None
    h = _mk(ev.HAAR, ev.BASIC, 1)
    a = _mk(ev.HAAR, ev.ALL, 1)
    l = _mk(ev.LBP, 0, 1)
    assert h.getNumFeatures() == 162336 and a.getNumFeatures() == 261600 and l.getNumFeatures() == 8464
    assert _mk(ev.HAAR, ev.CORE, 1).getNumFeatures() == 210400
    assert h.getMaxCatCount() == 0 and l.getMaxCatCount() == 256 and h.getFeatureSize() == 1
    assert _mk(ev.LBP, 0, 1, (75, 32)).getNumFeatures() == 152625  # res/README.md:41
    with pytest.raises(cc.CascadeError):
        _mk(ev.HAAR, ev.BASIC, 0)  # CV_Assert(_maxSampleCount > 0), features.cpp:75


def test_haar_constant_image_is_zero():  # test_features.cpp:252-275
    e = _mk(ev.HAAR, ev.BASIC, 1)
    e.setImage(np.full((24, 24), 128, np.uint8), 1, 0)
    v = e.calc_batch(0, e.getNumFeatures())
    assert not v.any() and e.getNumFeatures() > 0
    assert e(0, 0) == 0.0


def test_haar_step_edge_nonzero():  # test_features.cpp:277-298
    e = _mk(ev.HAAR, ev.BASIC, 1)
    img = np.zeros((24, 24), np.uint8)
    img[:, 12:] = 255
    e.setImage(img, 1, 0)
    assert e.calc_batch(0, e.getNumFeatures()).any()


def test_haar_all_mode_tilted_and_label():  # test_features.cpp:300-317
    e = _mk(ev.HAAR, ev.ALL, 1)
    e.setImage(np.full((24, 24), 64, np.uint8), 0, 0)
    assert e(0, 0) == 0.0 and e(e.getNumFeatures() - 1, 0) == 0.0
    assert e.getCls(0) == 0.0


def test_lbp_constant_image_is_255():  # test_features.cpp:319-340
    e = _mk(ev.LBP, 0, 1)
    e.setImage(np.full((24, 24), 50, np.uint8), 1, 0)
    assert (e.calc_batch(0, e.getNumFeatures()) == 255.0).all()


def test_lbp_isolates_samples_by_index():  # test_features.cpp:342-392
    e = _mk(ev.LBP, 0, 2)
    tex = np.zeros((24, 24), np.uint8)
    tex[12:, :] = 200
    e.setImage(np.full((24, 24), 80, np.uint8), 0, 0)
    e.setImage(tex, 1, 1)
    v = e.calc_batch(0, e.getNumFeatures())
    assert e(0, 0) == 255.0 and (v[:, 0] == 255).all() and (v[:, 1] < 255).any()
    assert e.getCls(0) == 0.0 and e.getCls(1) == 1.0


def test_setimage_asserts():  # features.cpp:85-87
    e = _mk(ev.LBP, 0, 2)
    with pytest.raises(cc.CascadeError):
        e.setImage(np.zeros((24, 25), np.uint8), 1, 0)
    with pytest.raises(cc.CascadeError):
        e.setImage(np.zeros((24, 24), np.uint8), 1, 2)


def test_feature_calc_kats():  # test_features.cpp:462-560: -3200, 0, -3600, tilted 32
    e = _mk(ev.HAAR, ev.BASIC, 2, (8, 8))
    img = np.zeros((8, 8), np.uint8)
    img[:, 4:] = 100
    e.setImage(img, 1, 0)
    e.setImage(np.full((8, 8), 42, np.uint8), 1, 1)
    v = e.calc_custom_haar([(False, [(0, 0, 4, 8, +1.0), (4, 0, 4, 8, -1.0)])], normalized=False)
    assert v[0, 0] == -3200.0 and v[0, 1] == 0.0
    e3 = _mk(ev.HAAR, ev.BASIC, 1, (9, 3))
    img = np.zeros((3, 9), np.uint8)
    img[:, 3:6] = 200
    e3.setImage(img, 1, 0)
    assert e3.calc_custom_haar([(False, [(0, 0, 9, 3, +1.0), (3, 0, 3, 3, -3.0)])])[0, 0] == -3600.0
    et = _mk(ev.HAAR, ev.ALL, 1, (16, 16))
    et.setImage(np.ones((16, 16), np.uint8), 1, 0)
    assert et.calc_custom_haar([(True, [(8, 2, 4, 4, +1.0)])])[0, 0] == 32.0


# ---- bulk parity with the oracle -----------------------------------------------------------------------------
def _samples(n, seed, W=24, H=24):
    rng = np.random.default_rng(seed)
    tmpl = rng.integers(0, 256, (H, W)).astype(np.float64)
    pos = np.clip(tmpl + rng.normal(0, 15, (n // 2, H, W)), 0, 255).astype(np.uint8)
    neg = rng.integers(0, 256, (n - n // 2, H, W), dtype=np.uint8)
    return np.concatenate([pos, neg])


@pytest.mark.parametrize("mode", [ev.BASIC, ev.ALL])
def test_setimages_integrals_and_normfactor(mode):
    imgs = _samples(37, 1)
    imgs[5] = 9  # flat sample: norm factor exactly 0
    e = _mk(ev.HAAR, mode, 40)
    e.setImages(imgs, np.arange(37) % 2, first_idx=2)
    s, t, nf = orc.set_images(imgs, want_tilted=(mode == ev.ALL))
    for i in (0, 5, 17, 36):
        gs, gt, gnf = e.get_sample(2 + i)
        assert (gs == s[i]).all()
        if mode == ev.ALL:
            assert (gt == t[i]).all()
        assert np.float32(gnf) == nf[i]
    assert e.getCls(2 + 3) == 1.0 and e.getCls(2 + 4) == 0.0


@pytest.mark.parametrize("mode", [ev.BASIC, ev.CORE, ev.ALL])
def test_haar_catalog_values_match_oracle(mode):
    imgs = _samples(50, 2)
    imgs[7] = 200
    e = _mk(ev.HAAR, mode, 50)
    e.setImages(imgs)
    feats = orc.haar_catalog(24, 24, mode)
    s, t, nf = orc.set_images(imgs, want_tilted=(mode == ev.ALL))
    n = e.getNumFeatures()
    assert n == len(feats)
    for (a, b) in [(0, 30000), (n // 2, n // 2 + 20000), (n - 25000, n)]:
        g = e.calc_batch(a, b)
        o = orc.haar_eval_batch(feats, a, b, s, t, nf, 24, 24)
        assert g.shape == o.shape
        assert (g.view(np.uint32) == o.view(np.uint32)).all(), f"{(g != o).sum()} of {g.size} differ"
    # geometry accessor (writeFeatures needs it) follows the catalog order
    for fi in (0, 1, n // 3, n - 1):
        r, w, tl = e.feature_geometry(fi)
        assert (r == feats["r"][fi]).all() and (w == feats["wt"][fi]).all() and tl == feats["tilted"][fi]


def test_haar_sample_index_gather_and_scalar_call():
    imgs = _samples(64, 3)
    e = _mk(ev.HAAR, ev.BASIC, 64)
    e.setImages(imgs)
    feats = orc.haar_catalog(24, 24, 0)
    s, t, nf = orc.set_images(imgs)
    idx = np.array([63, 0, 5, 5, 31, 2, 17], np.int32)
    g = e.calc_batch(1000, 1200, sample_idx=idx)
    o = orc.haar_eval_batch(feats, 1000, 1200, s, t, nf, 24, 24, sample_idx=idx)
    assert (g.view(np.uint32) == o.view(np.uint32)).all()
    assert np.float32(e(123456, 9)) == orc.haar_eval_batch(feats, 123456, 123457, s, t, nf, 24, 24, sample_idx=[9])[0, 0]


def test_lbp_catalog_bit_exact_on_barcode_samples(repo_root):
    samples = read_vec(os.path.join(repo_root, "tests", "golden", "barcode.vec"))  # 100 x 32 x 75
    e = _mk(ev.LBP, 0, 100, (75, 32))
    e.setImages(samples)
    rects = orc.lbp_catalog(75, 32)
    s, _, _ = orc.set_images(samples, want_norm=False)
    n = e.getNumFeatures()
    assert n == 152625
    for (a, b) in [(0, 20000), (70000, 90000), (n - 20000, n)]:
        assert (e.calc_batch(a, b) == orc.lbp_eval_batch(rects, a, b, s, 75, 32)).all()
    assert (e.feature_geometry(777)[0] == rects[777]).all()


def test_haar_basic_on_barcode_samples(repo_root):  # 75x32 window: 2 790 554 features (res/README.md:91)
    samples = read_vec(os.path.join(repo_root, "tests", "golden", "barcode.vec"))[:24]
    e = _mk(ev.HAAR, ev.BASIC, 24, (75, 32))
    e.setImages(samples)
    assert e.getNumFeatures() == 2790554
    feats = orc.haar_catalog(75, 32, 0)
    s, t, nf = orc.set_images(samples)
    for (a, b) in [(0, 40000), (2790554 - 40000, 2790554)]:
        g = e.calc_batch(a, b)
        o = orc.haar_eval_batch(feats, a, b, s, t, nf, 75, 32)
        assert (g.view(np.uint32) == o.view(np.uint32)).all()


@pytest.mark.parametrize("which", ["haar", "lbp"])
def test_training_side_cascade_predict(haar_xml, lbp_xml, which):
    """CvCascadeClassifier::predict over stored samples (negative-mining inner loop, cascadeclassifier.cpp:297-357)."""
    xml = haar_xml if which == "haar" else lbp_xml
    tm = np.load(os.path.join(os.path.dirname(xml), "face_template_24x24.npy"))
    rng = np.random.default_rng(5)
    near = np.clip(tm[None].astype(np.float64) + rng.normal(0, 6, (40, 24, 24)), 0, 255).astype(np.uint8)
    imgs = np.concatenate([near, _samples(160, 6), tm[None]])
    c = cc.CascadeClassifier(xml)
    o = orc.load_cascade_xml(xml)
    e = _mk(ev.HAAR if which == "haar" else ev.LBP, ev.BASIC, len(imgs))
    e.setImages(imgs)
    got = e.predict_cascade(c)
    s, t, nf = orc.set_images(imgs, want_norm=(which == "haar"))
    want = np.array([orc.train_predict(o, s, t, nf, i, 24, 24) for i in range(len(imgs))], np.uint8)
    assert (got == want).all()
    if which == "haar":
        assert got[-1] == 1 and 0 < got.sum() < len(imgs)


@pytest.mark.parametrize("which", ["haar", "haar_tilted", "lbp"])
def test_training_side_predict_with_trees(tmp_path, which):
    """CvCascadeBoostTree::predict on trees deeper than stumps (o_cvcascadeboosttree.cpp:16-39): ordered splits go left
    on `<=`, categorical on a set bit; stage passes on sum >= threshold - 1e-5f."""
    from tests import cascade_factory as cf
    from tests.util import frame_natural
    img = frame_natural(320, 240, 3)
    wins = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
    xml = cf.lbp_tree_cascade() if which == "lbp" else cf.haar_tree_cascade(wins, with_tilted=(which == "haar_tilted"))
    path = os.path.join(str(tmp_path), "c.xml")
    open(path, "w").write(xml)
    c = cc.CascadeClassifier(path)
    o = orc.load_cascade_xml(path)
    imgs = wins[:300]
    mode = ev.ALL if which == "haar_tilted" else ev.BASIC
    e = _mk(ev.LBP if which == "lbp" else ev.HAAR, mode, len(imgs))
    e.setImages(imgs)
    got = e.predict_cascade(c)
    s, t, nf = orc.set_images(imgs, want_tilted=(mode == ev.ALL), want_norm=(which != "lbp"))
    want = np.array([orc.train_predict(o, s, t, nf, i, 24, 24) for i in range(len(imgs))], np.uint8)
    assert (got == want).all() and 0 < got.sum() < len(imgs)


@pytest.mark.parametrize("ftype", [ev.HAAR, ev.LBP])
def test_sorted_index_precalc(ftype):
    """FeatureValAndIdxPrecalc (o_cvcascadeboosttraindata.cpp:522-562): values + per-feature argsort of the samples.
    Ties are ordered by sample index here (std::sort in the reference leaves them unspecified): compare with a stable
    argsort of the oracle's values."""
    imgs = _samples(300, 9)
    imgs[10] = imgs[11] = imgs[12]  # identical samples: tied values in every row
    imgs[20] = 7                    # flat: Haar values 0
    e = _mk(ftype, ev.BASIC, 300)
    e.setImages(imgs)
    n = e.getNumFeatures()
    a, b = n // 3, n // 3 + 700
    vals, idx = e.calc_batch_sorted(a, b)
    assert idx.dtype == np.uint16 and vals.shape == idx.shape == (700, 300)
    if ftype == ev.HAAR:
        s, t, nf = orc.set_images(imgs)
        want = orc.haar_eval_batch(orc.haar_catalog(24, 24, 0), a, b, s, t, nf, 24, 24)
    else:
        s, _, _ = orc.set_images(imgs, want_norm=False)
        want = orc.lbp_eval_batch(orc.lbp_catalog(24, 24), a, b, s, 24, 24)
    assert (vals.view(np.uint32) == want.view(np.uint32)).all()
    # -0.0 and +0.0 compare equal in the reference's comparator; normalise before the stable reference argsort
    ref = np.argsort(want + np.float32(0.0), axis=1, kind="stable")
    got_sorted = np.take_along_axis(want, idx.astype(np.int64), axis=1)
    assert (np.diff(got_sorted, axis=1) >= 0).all()          # non-decreasing values
    assert (np.sort(idx, axis=1) == np.arange(300)).all()    # a permutation of the samples
    same = (idx == ref).all(axis=1)
    # rows may differ from the stable order only where a row holds both -0.0 and +0.0 (radix order separates them)
    for r in np.nonzero(~same)[0]:
        assert (want[r] == 0).sum() >= 2 and np.signbit(want[r][want[r] == 0]).any()
    _, idx32 = e.calc_batch_sorted(a, a + 5, idx_bytes=4)
    assert idx32.dtype == np.int32 and (idx32 == idx[:5]).all()


def test_operator_call_is_safe_for_concurrent_callers():
    """The reference calls operator() from cv::parallel_for_ workers (o_cvcascadeboosttraindata.cpp:586-594): concurrent
    calc / calc_batch on one evaluator must give the same values as serial calls."""
    import threading
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (48, 24, 24), dtype=np.uint8)
    e = _mk(ev.HAAR, ev.BASIC, 48)
    e.setImages(imgs, np.zeros(48, np.uint8))
    want = e.calc_batch(1000, 1400)
    errors = []

    def work(t):
        try:
            for rep in range(6):
                f0 = 1000 + 50 * ((t + rep) % 8)
                got = e.calc_batch(f0, f0 + 50)
                assert (got.view(np.uint32) == want[f0 - 1000:f0 - 950].view(np.uint32)).all()
                fi, si = 1000 + (37 * t + rep) % 400, (11 * t + rep) % 48
                assert np.float32(e(fi, si)).view(np.uint32) == want[fi - 1000, si].view(np.uint32)
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_hoisted_division_is_the_division_operator():
    """operator() divides by the sample's norm factor (haarfeatures.h:108-112); the bulk kernel hoists the half of the
    IEEE division that depends on the divisor alone. 2^32 operand pairs of the evaluator's ranges, compared bit for bit
    with `a / b` on the device: no mismatch."""
    import ctypes as C
    bad = C.c_uint64(123)
    L.check(L.lib().cc_debug_division_check(0, 1 << 32, 20261004, C.byref(bad)))
    assert bad.value == 0


@pytest.mark.parametrize("ftype,mode", [(ev.HAAR, ev.ALL), (ev.LBP, 0)])
def test_feature_list_of_one_sample_matches_the_bulk_values(ftype, mode):
    """cc_eval_calc_list (the shape of the trainer's stage prediction on one freshly set window): any list of catalog
    features, repeated and unordered, for one stored sample == the same entries of the bulk evaluation == the oracle."""
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (6, 24, 24), dtype=np.uint8)
    imgs[4] = 200  # flat sample: norm factor 0 -> Haar values 0
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, mode), 6, (24, 24))
    e.setImages(imgs)
    F = e.getNumFeatures()
    lst = np.concatenate([rng.integers(0, F, 3000), [0, F - 1, 7, 7, 7]]).astype(np.int32)
    if ftype == ev.HAAR:
        s, t, nf = orc.set_images(imgs, want_tilted=True)
        want_all = orc.haar_eval_batch(orc.haar_catalog(24, 24, mode), 0, F, s, t, nf, 24, 24)
    else:
        s, t, nf = orc.set_images(imgs, want_norm=False)
        want_all = orc.lbp_eval_batch(orc.lbp_catalog(24, 24), 0, F, s, 24, 24)
    for si in (0, 3, 4, 5):
        got = e.calc_list(lst, si)
        assert (got.view(np.uint32) == want_all[lst, si].view(np.uint32)).all()
    assert len(e.calc_list([], 2)) == 0
    with pytest.raises(cc.CascadeError):
        e.calc_list([F], 0)
    with pytest.raises(cc.CascadeError):
        e.calc_list([0], 6)


@pytest.mark.parametrize("ftype,mode", [(ev.HAAR, ev.ALL), (ev.HAAR, ev.BASIC), (ev.LBP, 0)])
def test_host_mirror_of_the_last_set_window_equals_the_device(ftype, mode):
    """Round 4: cc_eval_set_image queues the window for the device and mirrors it on the host; cc_eval_calc /
    cc_eval_calc_list for THAT sample are answered from the mirror (no launch: the trainer's negative-mining loop makes
    these calls per window, cascadeclassifier.cpp:340-347). The mirror must give the device's bits for EVERY catalog
    feature (upright, tilted, LBP), its integrals / norm factor must be the device's, a flat window must give the nf == 0
    short-circuit, and a later image for the same or another sample must supersede it in the right order."""
    rng = np.random.default_rng(17)
    e = cc.CvFeatureEvaluator.create(ftype)
    e.init(cc.CvFeatureParams(ftype, mode), 5, (24, 24))
    F = e.getNumFeatures()
    allf = np.arange(F, dtype=np.int32)
    imgs = rng.integers(0, 256, (4, 24, 24), dtype=np.uint8)
    imgs[2] = 77  # flat: Haar norm factor 0
    for k, idx in enumerate((3, 3, 0, 4)):  # same slot twice (the loop's shape), then other slots
        padded = np.zeros((24, 40), np.uint8)
        padded[:, :24] = imgs[k]
        view = padded[:, :24]  # row stride 40: setImage takes strided windows
        L.check(L.lib().cc_eval_set_image(e._e, view.ctypes.data_as(C.c_void_p), 40, k & 1, idx))
        assert e.getCls(idx) == float(k & 1)
        host = e.calc_list(allf, idx)                 # mirror
        one = np.array([e(int(f), idx) for f in (0, F // 2, F - 1)], np.float32)
        dev = e.calc_batch(0, F, sample_idx=[idx])[:, 0]  # device (the queued image is flushed first)
        assert (host.view(np.uint32) == dev.view(np.uint32)).all(), (k, idx)
        assert (one.view(np.uint32) == dev[[0, F // 2, F - 1]].view(np.uint32)).all()
        if ftype == ev.HAAR and k == 2:
            assert (host == 0).all()
    # what the device holds after all that: slot 3 = image 1 (the later of the two), slot 0 = image 2, slot 4 = image 3
    if ftype == ev.HAAR:
        s, t, nf = orc.set_images(imgs, want_tilted=mode == ev.ALL)
    else:
        s, t, nf = orc.set_images(imgs, want_norm=False)
    for idx, k in ((3, 1), (0, 2), (4, 3)):
        got = e.get_sample(idx)
        assert (got[0] == s[k]).all()
        if ftype == ev.HAAR:
            assert got[2] == nf[k]
            if mode == ev.ALL:
                assert (got[1] == t[k]).all()
    # setImages over a mirrored slot drops the mirror
    e.setImage(imgs[0], 1, 2)
    e.setImages(imgs[1:2], first_idx=2)
    assert (e.calc_list(allf, 2).view(np.uint32) == e.calc_batch(0, F, sample_idx=[2])[:, 0].view(np.uint32)).all()
    assert (e.get_sample(2)[0] == s[1]).all()


def test_bulk_values_into_pitched_device_memory():
    """cc_eval_calc_batch_device: rows `pitch` floats apart in device memory; the padding is left untouched."""
    import torch
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (100, 24, 24), dtype=np.uint8)
    e = cc.CvFeatureEvaluator.create(0)
    e.init(cc.CvFeatureParams(0, 0), 100, (24, 24))
    e.setImages(imgs)
    want = e.calc_batch(1000, 1300)
    for pitch in (0, 100, 128, 133):
        p = pitch or 100
        out = torch.full((300, p), -7.0, dtype=torch.float32, device="cuda")
        e.calc_batch_device(1000, 1300, out.data_ptr(), n_samples=100, pitch=pitch)
        got = out.cpu().numpy()
        assert (got[:, :100].view(np.uint32) == want.view(np.uint32)).all()
        assert (got[:, 100:] == -7.0).all()
    with pytest.raises(cc.CascadeError):
        e.calc_batch_device(0, 10, torch.empty(10, 100, device="cuda").data_ptr(), n_samples=100, pitch=64)
