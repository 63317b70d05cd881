"""Run-time specialised cascade kernel (cc_detector_specialize, hiprtc): the first stages compiled into straight-line
code must not change a single bit. Per-window result codes, stage sums, visited flags, candidates and grouped rectangles
are compared with the CPU oracle AND with the table-driven kernel, for the synthetic stock-profile cascade (all
specialisation depths incl. the stump-split path), a cascade with tilted features and a non-24x24 window."""
import numpy as np
import pytest

import cascadeclassifier_amd as cc
from oracle import oracle as orc
from tests import cascade_factory as cf
from tests.util import frame_natural, frame_uniform

pytestmark = pytest.mark.gpu


def _same_as_oracle(p, o, img, sf):
    ref = orc.detect_raw(o, img, sf, nthreads=8, full=True)
    codes, sums, vis = p.debug_windows(img, sf)
    assert (codes == ref.codes).all(), f"{(codes != ref.codes).sum()} window results differ"
    assert (sums == ref.sums).all() and (vis == ref.visited).all()
    raw = p.detect_raw(img, sf)
    assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
    a, b = p.detectMultiScale(img, sf, 2), orc.detect_multiscale(o, img, sf, 2, nthreads=8)
    assert a.shape == b.shape and (a == b).all()
    return len(raw)


@pytest.mark.parametrize("k", [1, 4, 7])
def test_specialised_stock_profile_cascade(haar_xml, k):
    o = orc.load_cascade_xml(haar_xml)
    p = cc.CascadeClassifier(haar_xml)
    assert p.specialized_stages() == 0
    got = p.specialize(k)
    assert 1 <= got <= k and p.specialized_stages() == got
    img = frame_natural(640, 360, 11)
    n = _same_as_oracle(p, o, img, 1.1)
    n += _same_as_oracle(p, o, frame_uniform(300, 200, 12), 1.25)
    assert n > 0
    # batch path (several passes, both streams) against the table-driven kernel
    frames = np.stack([frame_natural(480, 270, 20 + i) for i in range(6)])
    spec = p.detect_batch(frames, 1.1, 3)
    p.specialize(0)
    assert p.specialized_stages() == 0
    plain = p.detect_batch(frames, 1.1, 3)
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(spec, plain))


def test_specialised_mixed_fixed_point_and_double_stages(haar_xml, tmp_path):
    """The generated stages accumulate votes as int32 multiples of the stage's leaf quantum where the sums fit and as
    doubles otherwise. A tiny leaf (quantum 2^-45) in stages 1 and 3 of the stock-profile cascade pushes those two out of
    the integer form while their neighbours keep it: results must still match the oracle bit for bit."""
    import re
    text = open(haar_xml).read()
    sizes = [int(v) for v in re.findall(r"<maxWeakCount>(\d+)</maxWeakCount>", text[text.index("<stages>"):])]
    leaves = list(re.finditer(r"<leafValues>\s*(\S+)\s+(\S+)</leafValues>", text))
    assert len(leaves) == sum(sizes)
    edits = {sum(sizes[:1]) + 2: ("3.1e-07", "-3.1e-07"), sum(sizes[:3]) + 5: ("-2.9e-07", "4.4e-07")}
    out, last = [], 0
    for i, mt in enumerate(leaves):
        if i in edits:
            out.append(text[last:mt.start()] + "<leafValues>%s %s</leafValues>" % edits[i])
            last = mt.end()
    text = "".join(out) + text[last:]
    path = str(tmp_path / "mixed.xml")
    open(path, "w").write(text)
    o = orc.load_cascade_xml(path)
    p = cc.CascadeClassifier(path)
    assert p.specialize(7) == 7
    n = _same_as_oracle(p, o, frame_natural(640, 360, 11), 1.1)
    n += _same_as_oracle(p, o, frame_uniform(300, 200, 12), 1.25)
    assert n > 0


def test_specialised_tilted_and_other_windows(tmp_path):
    img = frame_natural(320, 240, 3)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
    xml = cf.tilted_stump_cascade(cal)
    path = str(tmp_path / "t.xml")
    open(path, "w").write(xml)
    p = cc.CascadeClassifier(path)
    assert p.specialize(8) >= 1
    _same_as_oracle(p, orc.load_cascade_xml(path), frame_natural(333, 127, 6), 1.2)


def test_specialised_lbp_cascade(lbp_xml):
    """The stock LBP cascade (20 stages, 139 stumps) compiled whole: bit-exact like the table-driven kernel."""
    o = orc.load_cascade_xml(lbp_xml)
    p = cc.CascadeClassifier(lbp_xml)
    assert p.specialize(20) == 20
    n = _same_as_oracle(p, o, frame_natural(640, 360, 11), 1.1)
    n += _same_as_oracle(p, o, frame_uniform(300, 200, 12), 1.25)
    frames = np.stack([frame_natural(480, 270, 40 + i) for i in range(5)])
    spec = p.detect_batch(frames, 1.1, 2)
    p.specialize(0)
    plain = p.detect_batch(frames, 1.1, 2)
    assert all(a.shape == b.shape and (a == b).all() for a, b in zip(spec, plain))


def test_specialise_refuses_what_it_does_not_cover(tmp_path):
    img = frame_natural(320, 240, 3)
    cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
    trees = cc.CascadeClassifier()
    assert trees.load_from_string(cf.haar_tree_cascade(cal, with_tilted=False))
    with pytest.raises(cc.CascadeError, match="stump cascades only"):
        trees.specialize(2)


def test_background_specialisation_switches_over(haar_xml, monkeypatch):
    """cc_detector_specialize_async / CCAMD_AUTO_SPECIALIZE: detection runs on the table-driven kernel while a host thread
    compiles, and a later call picks the module up; results identical before and after."""
    import time
    img = frame_natural(400, 300, 91)
    want = orc.detect_multiscale(orc.load_cascade_xml(haar_xml), img, 1.1, 2, nthreads=8)
    p = cc.CascadeClassifier(haar_xml)
    p.specialize_async(2)
    seen_plain = p.specialized_stages() == 0
    deadline = time.time() + 120
    while p.specialized_stages() == 0 and time.time() < deadline:
        got = p.detectMultiScale(img, 1.1, 2)  # keeps working during the build
        assert got.shape == want.shape and (got == want).all()
        time.sleep(0.05)
    assert seen_plain and p.specialized_stages() == 2
    got = p.detectMultiScale(img, 1.1, 2)
    assert got.shape == want.shape and (got == want).all()
    # the environment switch does the same for a detector created without any call
    monkeypatch.setenv("CCAMD_AUTO_SPECIALIZE", "2")
    q = cc.CascadeClassifier(haar_xml)
    deadline = time.time() + 120
    while q.specialized_stages() == 0 and time.time() < deadline:
        got = q.detectMultiScale(img, 1.1, 2)
        assert got.shape == want.shape and (got == want).all()
        time.sleep(0.02)
    assert q.specialized_stages() == 2
    del q  # destroying a detector joins its build thread
    r = cc.CascadeClassifier(haar_xml)
    r.specialize_async(3)
    del r  # ... also while the build is still running
