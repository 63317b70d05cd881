"""File formats either side of the hot path (host only): cascade.xml round trip through the product's writer and
reader, and the .vec sample format against the reference's own fixture (traincascade/res/barcode.vec)."""
import hashlib
import os

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import detector as det
from tests import cascade_factory as cf
from tests.util import frame_natural, read_vec


def _models_equal(a, b):
    assert a.info == b.info
    for f in ("stage_first", "stage_ntrees", "stage_threshold", "rects"):
        assert (getattr(a, f) == getattr(b, f)).all(), f
    for f in ("stump_feature", "stump_threshold", "stump_left", "stump_right", "stump_subsets", "weights", "tilted"):
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None) == (y is None) and (x is None or (x.view(np.uint32) == y.view(np.uint32)).all()), f


def _calib():
    img = frame_natural(320, 240, 3)
    return np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])


@pytest.mark.parametrize("which", ["haar", "lbp", "tilted", "haar_trees", "lbp_trees"])
def test_cascade_xml_round_trip(tmp_path, haar_xml, lbp_xml, which):
    src = cc.CascadeClassifier()
    if which == "haar":
        assert src.load(haar_xml)
    elif which == "lbp":
        assert src.load(lbp_xml)
    else:
        xml = {"tilted": lambda: cf.tilted_stump_cascade(_calib()), "haar_trees": lambda: cf.haar_tree_cascade(_calib(), with_tilted=True),
               "lbp_trees": cf.lbp_tree_cascade}[which]()
        assert src.load_from_string(xml), getattr(src, "load_error", "")
    out = os.path.join(str(tmp_path), "saved.xml")
    src.save(out)
    back = cc.CascadeClassifier(out)
    assert not back.empty(), getattr(back, "load_error", "")
    if src.info()["max_nodes_per_tree"] == 1:
        _models_equal(src.model(), back.model())
    else:
        assert src.info() == back.info()
        a, b = src.model(), back.model()
        assert (a.stage_threshold.view(np.uint32) == b.stage_threshold.view(np.uint32)).all() and (a.rects == b.rects).all()
    # and the independent (ElementTree) reader of the oracle accepts the written file
    from oracle import oracle as orc
    o = orc.load_cascade_xml(out)
    assert o.nstages == src.info()["n_stages"] and o.nfeatures == src.info()["n_features"]
    again = os.path.join(str(tmp_path), "saved2.xml")
    back.save(again)
    assert open(out, "rb").read() == open(again, "rb").read()  # writer is a fixed point


def test_vec_reader_on_the_reference_fixture(repo_root):
    path = os.path.join(repo_root, "tests", "golden", "barcode.vec")
    got = det.vec_read(path)
    assert got.shape == (100, 2400)
    assert (got.reshape(100, 32, 75) == read_vec(path)).all()
    assert det.vec_read(path, max_samples=3).shape == (3, 2400)


def test_vec_writer_reproduces_the_reference_file_byte_for_byte(repo_root, tmp_path):
    path = os.path.join(repo_root, "tests", "golden", "barcode.vec")
    out = os.path.join(str(tmp_path), "copy.vec")
    det.vec_write(out, det.vec_read(path), 75, 32)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == "f76315c9351c82aeaabd66b066c203f358a11d1e69c1cc5906a0373320fe76db"


def test_vec_errors(tmp_path):
    with pytest.raises(cc.CascadeError):
        det.vec_read("/nonexistent.vec")
    bad = os.path.join(str(tmp_path), "bad.vec")
    open(bad, "wb").write(bytes([5, 0, 0, 0, 16, 0, 0, 0, 0, 0, 0, 0, 0]) + b"abc")  # header promises 5 samples of 16 px
    with pytest.raises(cc.CascadeError):
        det.vec_read(bad)
