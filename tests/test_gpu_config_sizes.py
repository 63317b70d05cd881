"""BASELINE.json's configurations at their stated sizes (the other GPU tests keep inputs small):
  configs[2]  stock LBP cascade on a 1920x1080 frame, bit-exact integer path;
  configs[3]  a 64-frame Full-HD batch on one GPU (one rank's share of the 512-frame job), specialised kernel;
  configs[4]  training evaluation at 10 000 + 10 000 samples of 24x24, Haar BASIC (162 336 features): bulk operator(),
              the sorted-index precalc with 16-bit indices, the split search with the per-sample table in LDS and in
              global memory, and 20 481 samples (one past the LDS-table limit, cc_split.hip).
Everything is compared with the CPU oracle on the same seeded inputs."""
import os

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import evaluator as ev
from oracle import oracle as orc
from tests.test_gpu_split import _node_value, _weights
from tests.util import frame_natural

pytestmark = pytest.mark.gpu


def _faces(img, seed, ks):
    tm = np.load(os.path.join(os.path.dirname(__file__), "..", "data", "face_template_24x24.npy"))
    rng = np.random.default_rng(seed)
    h, w = img.shape
    for k in ks:
        s = int(24 * k)
        y, x = int(rng.integers(0, h - s)), int(rng.integers(0, w - s))
        img[y:y + s, x:x + s] = orc.resize_linear_exact(tm, s, s)
    return img


def test_config3_lbp_full_hd_bit_exact(lbp_xml):
    img = _faces(frame_natural(1920, 1080, 3), 3, (1.0, 2.2, 4.0, 7.5))
    o = orc.load_cascade_xml(lbp_xml)
    p = cc.CascadeClassifier(lbp_xml)
    ref = orc.detect_raw(o, img, 1.1, nthreads=16, full=True)
    for spec in (0, 20):  # table-driven kernel, then all 20 stages specialised
        if spec:
            try:
                p.specialize(spec)
            except cc.CascadeError:
                pytest.skip("libhiprtc not available")
        codes, sums, vis = p.debug_windows(img, 1.1)
        assert len(codes) == 4514050 == ref.n_grid_windows
        assert (codes == ref.codes).all() and (vis == ref.visited).all() and (sums == ref.sums).all()
        raw = p.detect_raw(img, 1.1)
        assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
        a = p.detectMultiScale(img, 1.1, 3)
        b = orc.detect_multiscale(o, img, 1.1, 3, nthreads=16)
        assert a.shape == b.shape and (a == b).all()


def test_config4_one_ranks_share_64_full_hd_frames(haar_xml):
    """What bench.py times: detect_batch over 64 resident Full-HD frames with the first 7 stages specialised."""
    import torch
    frames = np.stack([_faces(frame_natural(1920, 1080, 200 + i), 200 + i, (1.0, 1.6, 2.7, 4.5, 8.0)) for i in range(64)])
    t = torch.from_numpy(frames).cuda()
    p = cc.CascadeClassifier(haar_xml, max_batch=64)
    try:
        p.specialize(7)
    except cc.CascadeError:
        pass  # no libhiprtc: the table-driven kernel is checked instead
    got = p.detect_batch(None, 1.1, 3, device_ptr=t.data_ptr(), shape=tuple(frames.shape))
    assert len(got) == 64
    o = orc.load_cascade_xml(haar_xml)
    for i in (0, 31, 63):
        want = orc.detect_multiscale(o, frames[i], 1.1, 3, nthreads=16)
        assert got[i].shape == want.shape and (got[i] == want).all(), i
        assert len(want) >= 3
    again = p.detect_batch(None, 1.1, 3, device_ptr=t.data_ptr(), shape=tuple(frames.shape))
    assert all((a == b).all() for a, b in zip(got, again))  # deterministic


def _config5_samples(n, seed=7):
    """positives = one template + N(0, 15^2) noise, negatives = uniform noise, 24x24, seed 7."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:24, 0:24]
    tmpl = 128 + 60 * np.sin(xx / 24 * 3.1) * np.cos(yy / 24 * 2.3)
    npos = n // 2
    pos = np.clip(tmpl[None] + rng.normal(0, 15, (npos, 24, 24)), 0, 255).astype(np.uint8)
    neg = rng.integers(0, 256, (n - npos, 24, 24), dtype=np.uint8)
    return np.concatenate([pos, neg]), np.concatenate([np.ones(npos, np.uint8), np.zeros(n - npos, np.uint8)])


@pytest.fixture(scope="module")
def config5():
    n = 20000
    imgs, labels = _config5_samples(n)
    e = cc.CvFeatureEvaluator.create(ev.HAAR)
    e.init(cc.CvFeatureParams(ev.HAAR, ev.BASIC), n, (24, 24))
    e.setImages(imgs, labels)
    s, t, nf = orc.set_images(imgs)
    return e, imgs, labels, (s, t, nf), orc.haar_catalog(24, 24, 0)


def test_config5_bulk_evaluation_20000_samples(config5):
    e, imgs, labels, (s, t, nf), cat = config5
    assert e.getNumFeatures() == 162336
    for lo in (0, 81000, 162336 - 4096):  # three slices of the catalog: first, middle, last
        got = e.calc_batch(lo, lo + 4096)
        want = orc.haar_eval_batch(cat, lo, lo + 4096, s, t, nf, 24, 24)
        assert got.shape == (4096, 20000)
        assert (got.view(np.uint32) == want.view(np.uint32)).all()


def test_config5_sorted_index_precalc_u16(config5):
    """precalculate's buf rows (o_cvcascadeboosttraindata.cpp:490-556) at 20 000 samples: 16-bit indices (is_buf_16u)."""
    e, imgs, labels, (s, t, nf), cat = config5
    lo, hi = 40000, 40512
    vals, idx = e.calc_batch_sorted(lo, hi)
    assert idx.dtype == np.uint16 and idx.shape == (512, 20000)
    want = orc.haar_eval_batch(cat, lo, hi, s, t, nf, 24, 24)
    assert (vals.view(np.uint32) == want.view(np.uint32)).all()
    order = np.argsort(want, axis=1, kind="stable")  # equal values keep sample order
    assert (idx.astype(np.int64) == order).all()


def _split_case(e, cat, ints, labels, lo, hi, n, seed):
    s, t, nf = ints
    lab = labels[:n].astype(np.int32)
    w = _weights(n, lab, seed, False)
    resp = (lab * 2 - 1).astype(np.float32)
    nv = _node_value(w, resp)
    got, gq, gpt = e.find_best_split(w, responses=resp, node_value=nv, per_var=True)
    vals = orc.haar_eval_batch(cat, lo, hi, s[:n], t, nf[:n], 24, 24)
    want, wq, wpt = orc.find_best_split(vals, w, responses=resp, node_value=nv, per_feature=True)
    assert (gpt == wpt).all()
    assert (np.where(gpt >= 0, gq.astype(np.float32), np.float32(-1)).view(np.uint32) == wq.view(np.uint32)).all()
    assert got["found"] and bool(want["found"])
    assert got["var_idx"] == lo + want["var_idx"] and got["quality"] == want["quality"]
    assert got["ord_c"] == want["ord_c"] and got["split_point"] == want["split_point"]


def test_config5_split_search_20000_samples_lds_and_global_table(config5, monkeypatch):
    e, imgs, labels, ints, cat = config5
    lo, hi = 60000, 60000 + 8192
    e.presort(20000, lo, hi)
    _split_case(e, cat, ints, labels, lo, hi, 20000, 5)  # per-sample table in LDS (n <= 20 480)
    monkeypatch.setenv("CCAMD_SPLIT_GLOBAL_TABLE", "1")
    e.presort(20000, lo, hi)
    _split_case(e, cat, ints, labels, lo, hi, 20000, 5)  # same search with the table gathered from global memory


def test_config5_one_past_the_lds_table_limit():
    """20 481 samples: the per-sample table no longer fits the block's LDS (cc_split.hip) and the search switches to the
    global-memory table by itself."""
    n = 20481
    imgs, labels = _config5_samples(n, seed=8)
    e = cc.CvFeatureEvaluator.create(ev.HAAR)
    e.init(cc.CvFeatureParams(ev.HAAR, ev.BASIC), n, (24, 24))
    e.setImages(imgs, labels)
    lo, hi = 1000, 1000 + 1024
    e.presort(n, lo, hi)
    s, t, nf = orc.set_images(imgs)
    _split_case(e, orc.haar_catalog(24, 24, 0), (s, t, nf), labels, lo, hi, n, 9)
