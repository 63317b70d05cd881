"""Detection with the less common cascade shapes: tilted (45 degree) Haar features and trees deeper than stumps.
Same bar as tests/test_gpu_detect.py: per-window result codes, stage sums, visited flags, candidates and grouped
rectangles identical to the CPU oracle."""
import os

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import detector as det
from oracle import oracle as orc
from tests import cascade_factory as cf
from tests.util import frame_natural, frame_uniform

pytestmark = pytest.mark.gpu


def _calib_windows():
    img = frame_natural(320, 240, 3)
    return np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])


def _check(xml_text, tmp_path, images, sfs=(1.1, 1.3)):
    path = os.path.join(tmp_path, "cascade.xml")
    open(path, "w").write(xml_text)
    o = orc.load_cascade_xml(path)
    p = cc.CascadeClassifier()
    assert p.load_from_string(xml_text), getattr(p, "load_error", "")
    n_cand = 0
    for img in images:
        for sf in sfs:
            ref = orc.detect_raw(o, img, sf, nthreads=8, full=True)
            codes, sums, vis = p.debug_windows(img, sf)
            assert (codes == ref.codes).all(), f"{(codes != ref.codes).sum()} window results differ"
            assert (sums == ref.sums).all()
            assert (vis == ref.visited).all()
            raw = p.detect_raw(img, sf)
            assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
            for mn in (0, 2):
                a, b = p.detectMultiScale(img, sf, mn), orc.detect_multiscale(o, img, sf, mn, nthreads=8)
                assert a.shape == b.shape and (a == b).all()
            n_cand += len(raw)
    return n_cand


@pytest.mark.parametrize("w,h", [(61, 47), (320, 240), (1000, 37)])
def test_tilted_integral_of_whole_images(w, h):
    img = frame_uniform(w, h, 9)
    assert (det.integral(img, tilted=True)["tilted"] == orc.integral(img, tilted=True)["tilted"]).all()


def test_tilted_haar_cascade(tmp_path):
    xml = cf.tilted_stump_cascade(_calib_windows())
    assert cc.CascadeClassifier().load_from_string(xml)
    n = _check(xml, str(tmp_path), [frame_natural(200, 150, 5), frame_natural(333, 127, 6), frame_uniform(120, 90, 7)])
    assert n > 0


@pytest.mark.parametrize("with_tilted", [False, True])
def test_haar_trees_deeper_than_stumps(tmp_path, with_tilted):
    xml = cf.haar_tree_cascade(_calib_windows(), with_tilted=with_tilted)
    p = cc.CascadeClassifier()
    assert p.load_from_string(xml) and p.info()["max_nodes_per_tree"] == 4
    n = _check(xml, str(tmp_path), [frame_natural(200, 150, 15), frame_natural(97, 211, 16)])
    assert n > 0


def test_lbp_trees_deeper_than_stumps(tmp_path):
    xml = cf.lbp_tree_cascade()
    n = _check(xml, str(tmp_path), [frame_natural(200, 150, 25), frame_uniform(150, 100, 26)])
    assert n > 0


def test_cyclic_tree_is_refused():
    xml = cf.lbp_tree_cascade().replace("<internalNodes>1 2 ", "<internalNodes>0 2 ", 1) if False else None
    bad = cf.haar_xml(orc.haar_catalog(24, 24, 0)[:2], [(np.float32(0), [([(1, -1, 0, np.float32(0)), (1, -2, 1, np.float32(0))], [0.1, 0.2, 0.3])])])
    p = cc.CascadeClassifier()
    assert not p.load_from_string(bad) and "cycle" in p.load_error


@pytest.mark.parametrize("W,H", [(75, 32), (96, 96), (20, 20), (31, 57)])
def test_windows_other_than_24x24(tmp_path, W, H):
    """The LDS tile geometry follows the cascade's window size (75x32 is the reference's barcode window,
    traincascade/res/README.md; 96x96 needs more than 64 KiB of LDS per tile)."""
    xml = cf.lbp_stump_cascade(W, H)
    n = _check(xml, str(tmp_path), [frame_natural(400, 300, 51), frame_natural(W + 3, H + 40, 52)], sfs=(1.1, 1.5))
    assert n > 0


@pytest.mark.parametrize("W,H", [(20, 20), (31, 57), (75, 32), (96, 96), (25, 24)])
def test_specialised_windows_other_than_24x24(tmp_path, W, H):
    """The same cascades with EVERY stage compiled (cc_detector_specialize): an LBP cascade compiled whole takes the 16-bit
    STEP-2 tile (TileGeom16) when all its cells sum below 2^16 -- 20x20, 31x57, 25x24 and 75x32 do, 96x96 has cells of up
    to 32x32 pixels and keeps the 32-bit tile -- so tile geometry, row pitch and bank skew of both layouts are exercised for
    even and odd window sizes."""
    xml = cf.lbp_stump_cascade(W, H)
    path = os.path.join(str(tmp_path), "cascade.xml")
    open(path, "w").write(xml)
    o = orc.load_cascade_xml(path)
    p = cc.CascadeClassifier(path)
    assert p.specialize(3) == 3
    n = 0
    for img, sf in ((frame_natural(400, 300, 51), 1.1), (frame_natural(W + 3, H + 40, 52), 1.5), (frame_uniform(333, 127, 53), 1.25)):
        ref = orc.detect_raw(o, img, sf, nthreads=8, full=True)
        codes, sums, vis = p.debug_windows(img, sf)
        assert (codes == ref.codes).all() and (sums == ref.sums).all() and (vis == ref.visited).all()
        raw = p.detect_raw(img, sf)
        assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
        n += len(raw)
    assert n > 0


@pytest.mark.parametrize("which", ["haar", "lbp", "haar_specialised"])
def test_output_reject_levels_overload(which, haar_xml, lbp_xml):
    """detectMultiScale(objects, rejectLevels, levelWeights, ..., outputRejectLevels=true): rectangles, levels (= number of
    stages for every accepted window) and weights (the last stage's sum, the largest one per group) against the oracle."""
    xml = lbp_xml if which == "lbp" else haar_xml
    o = orc.load_cascade_xml(xml)
    p = cc.CascadeClassifier(xml)
    if which == "haar_specialised":
        p.specialize(7)
    total = 0
    for img, sf, mn in ((frame_natural(640, 360, 11), 1.1, 2), (frame_natural(400, 300, 12), 1.2, 0), (frame_uniform(200, 150, 13), 1.1, 1)):
        rects, levels, weights = p.detectMultiScale3(img, sf, mn, outputRejectLevels=True)
        wr, wl, ww = orc.detect_multiscale_levels(o, img, sf, mn, nthreads=8)
        key_g = np.lexsort((weights, rects[:, 3], rects[:, 2], rects[:, 1], rects[:, 0])) if len(rects) else np.zeros(0, int)
        key_w = np.lexsort((ww, wr[:, 3], wr[:, 2], wr[:, 1], wr[:, 0])) if len(wr) else np.zeros(0, int)
        assert rects.shape == wr.shape and (rects[key_g] == wr[key_w]).all()
        assert (levels[key_g] == wl[key_w]).all() and (levels == o.nstages).all()
        assert (weights[key_g] == ww[key_w]).all()  # bit-exact stage sums
        plain = p.detectMultiScale(img, sf, mn)
        assert plain.shape == rects.shape  # same rectangles as the ordinary overload
        total += len(rects)
    assert total > 0
