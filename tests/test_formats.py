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


def _legacy_walk(tree_el):
    """(feature geometry, threshold, left, right) per node of one legacy <trees>/<_> element; children are
    ('node', k) or ('val', v)."""
    nodes = []
    for n in [e for e in tree_el if e.tag == "_"]:
        feat = n.find("feature")
        rects = [[float(v) for v in r.text.split()] for r in feat.find("rects") if r.tag == "_"]
        side = []
        for name in ("left", "right"):
            if n.find(name + "_node") is not None:
                side.append(("node", int(n.findtext(name + "_node"))))
            else:
                side.append(("val", float(n.findtext(name + "_val"))))
        nodes.append((rects, int(feat.findtext("tilted")), np.float32(float(n.findtext("threshold"))), side[0], side[1]))
    return nodes


@pytest.mark.parametrize("which", ["haar", "haar_trees"])
def test_legacy_base_format_writer(tmp_path, haar_xml, which):
    """cc_cascade_save_xml_legacy writes the layout of CvCascadeClassifier::save(filename, baseFormat=true)
    (cascadeclassifier.cpp:457-531): every stage, tree, node, feature and value of the model is found again by an
    independent ElementTree walk, with the breadth-first node numbering of the reference's queue."""
    import xml.etree.ElementTree as ET

    from oracle import oracle as orc
    src = cc.CascadeClassifier()
    if which == "haar":
        assert src.load(haar_xml)
        new_xml = haar_xml
    else:
        new_xml = os.path.join(str(tmp_path), "trees.xml")
        open(new_xml, "w").write(cf.haar_tree_cascade(_calib(), with_tilted=True))
        assert src.load(new_xml)
    out = os.path.join(str(tmp_path), "legacy.xml")
    src.save(out, baseFormat=True)
    root = ET.parse(out).getroot()
    casc = list(root)[0]
    assert casc.tag == "cascade" and casc.attrib["type_id"] == "opencv-haar-classifier"
    assert [int(v) for v in casc.findtext("size").split()] == [src.info()["win_w"], src.info()["win_h"]]
    o = orc.load_cascade_xml(new_xml)
    raw_thr = [np.float32(float(s.findtext("stageThreshold"))) for s in ET.parse(new_xml).getroot().iter("_") if s.find("stageThreshold") is not None]
    stages = [s for s in casc.find("stages") if s.tag == "_"]
    assert len(stages) == o.nstages
    tree = 0
    node0 = 0
    leaf0 = 0
    for si, st in enumerate(stages):
        assert np.float32(float(st.findtext("stage_threshold"))) == raw_thr[si]
        assert int(st.findtext("parent")) == si - 1 and int(st.findtext("next")) == -1
        trees = [t for t in st.find("trees") if t.tag == "_"]
        assert len(trees) == o.stage_ntrees[si]
        for t in trees:
            nodes = _legacy_walk(t)
            nn = int(o.tree_nnodes[tree])
            assert len(nodes) == nn
            # walk both representations from the root in the writer's breadth-first order
            order = [0]
            next_idx = 0
            for q, (rects, tilted, thr, left, right) in enumerate(nodes):
                k = node0 + order[q]
                f = int(o.node_feature[k])
                want = [[*o.haar["r"][f, j], o.haar["wt"][f, j]] for j in range(3) if j == 0 or o.haar["r"][f, j, 2] != 0 or o.haar["wt"][f, j] != 0]
                assert rects == [[float(v) for v in r] for r in want] and tilted == int(o.haar["tilted"][f])
                assert thr == o.node_threshold[k]
                for (kind, v), child in ((left, int(o.node_left[k])), (right, int(o.node_right[k]))):
                    if child > 0:
                        next_idx += 1
                        order.append(child)
                        assert (kind, v) == ("node", next_idx)
                    else:
                        assert kind == "val" and np.float32(v) == o.leaves[leaf0 - child]
            tree += 1
            node0 += nn
            leaf0 += nn + 1
    assert tree == len(o.tree_nnodes)
    # the product's reader refuses the legacy layout (as documented), and LBP cascades cannot be written in it
    assert not cc.CascadeClassifier().load(out)


def test_legacy_base_format_is_haar_only(tmp_path, lbp_xml):
    with pytest.raises(cc.CascadeError, match="old file format is used for Haar-like features only"):
        cc.CascadeClassifier(lbp_xml).save(os.path.join(str(tmp_path), "x.xml"), baseFormat=True)


def test_numbers_do_not_follow_the_host_programs_locale(tmp_path):
    """A host program that called setlocale(LC_ALL, "") under a comma-decimal locale must still get cascade XML with '.'
    decimals and must still parse it (and generated kernel source with valid float literals). Needs such a locale on
    the machine; skipped where none is installed."""
    import locale
    import ctypes as C

    from cascadeclassifier_amd import _lib as L
    old = locale.setlocale(locale.LC_NUMERIC)
    chosen = None
    for name in ("de_DE.UTF-8", "de_DE.utf8", "fr_FR.UTF-8", "fr_FR.utf8", "ru_RU.UTF-8", "nl_NL.UTF-8"):
        try:
            locale.setlocale(locale.LC_NUMERIC, name)
            chosen = name
            break
        except locale.Error:
            continue
    if chosen is None:
        pytest.skip("no comma-decimal locale installed")
    try:
        assert locale.localeconv()["decimal_point"] == ","
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        src = os.path.join(root, "data", "haarcascade_frontalface_synthetic.xml")
        c = C.c_void_p()
        L.check(L.lib().cc_cascade_load_xml(src.encode(), C.byref(c)))
        out = os.path.join(str(tmp_path), "again.xml")
        L.check(L.lib().cc_cascade_save_xml(c, out.encode()))
        text = open(out).read()
        assert "," not in text.split("<stages>")[1].split("</stages>")[0]
        c2 = C.c_void_p()
        L.check(L.lib().cc_cascade_load_xml(out.encode(), C.byref(c2)))
        n = C.c_size_t(0)
        L.check(L.lib().cc_cascade_compile_specialized(c2, 2, b"gfx950", C.byref(n)))  # "%a" literals compile
        assert n.value > 0
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)
