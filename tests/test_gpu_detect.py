"""GPU parity tests of the detection path: every stage of the HIP pipeline is compared with the CPU oracle through the
C ABI on seeded inputs. Integer / index work (resize, integrals, LBP codes, result codes, visited flags, rectangles) is
BIT-EXACT; Haar stage sums are compared exactly too (tolerance 0.0: the kernels reproduce the CPU operation order with
FMA contraction off, float feature values and a double accumulator), see HAAR_SUM_TOL."""
import os

import ctypes as C

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import _lib as L
from cascadeclassifier_amd import detector as det
from oracle import oracle as orc
from tests.util import frame_natural, frame_uniform

pytestmark = pytest.mark.gpu

HAAR_SUM_TOL = 0.0  # absolute tolerance on Haar stage sums (north_star allows a stated tolerance; we hold exact)


def _faces(img, seed, ks=(1.0, 1.7, 2.6, 4.0)):
    tm = np.load(os.path.join(os.path.dirname(__file__), "..", "data", "face_template_24x24.npy"))
    rng = np.random.default_rng(seed)
    out = img.copy()
    h, w = out.shape
    for k in ks:
        s = int(24 * k)
        if s >= min(h, w):
            continue
        u = orc.resize_linear_exact(tm, s, s)
        y, x = int(rng.integers(0, h - s)), int(rng.integers(0, w - s))
        out[y:y + s, x:x + s] = u
    return out


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 582, 436), (640, 480, 160, 120), (1920, 1080, 1745, 982), (101, 57, 33, 19),
                                         (64, 64, 64, 64), (50, 40, 49, 39), (333, 127, 7, 3), (24, 24, 1, 1)])
def test_resize_bit_exact(sw, sh, dw, dh):
    img = frame_uniform(sw, sh, 11)
    assert (det.resize_linear_exact(img, dw, dh) == orc.resize_linear_exact(img, dw, dh)).all()


@pytest.mark.parametrize("w,h", [(24, 24), (25, 31), (640, 480), (1920, 1080), (257, 3), (3, 257), (1000, 1)])
def test_integral_bit_exact(w, h):
    img = frame_uniform(w, h, 5)
    if w * h > 100000:
        img[: h // 2] = 255  # force the squared sum past 2^32
    g = det.integral(img, sqsum=True)
    o = orc.integral(img, sqsum_i32=True)
    assert (g["sum"] == o["sum"]).all()
    assert (g["sqsum"] == o["sqsum_i32"]).all()


def test_integral_tilted_bit_exact():
    img = frame_uniform(61, 47, 9)
    g = det.integral(img, tilted=True)
    assert (g["tilted"] == orc.integral(img, tilted=True)["tilted"]).all()


def _compare_windows(xml, img, sf, haar, min_size=None, max_size=None):
    o = orc.load_cascade_xml(xml)
    p = cc.CascadeClassifier(xml)
    ref = orc.detect_raw(o, img, sf, min_size or (0, 0), max_size or (0, 0), nthreads=8, full=True)
    codes, sums, vis = p.debug_windows(img, sf, min_size, max_size)
    assert len(codes) == ref.n_grid_windows
    assert (codes == ref.codes).all(), f"{(codes != ref.codes).sum()} window results differ"
    if haar:
        assert np.max(np.abs(sums - ref.sums), initial=0.0) <= HAAR_SUM_TOL
    else:
        assert (sums == ref.sums).all()
    assert (vis == ref.visited).all()
    raw = p.detect_raw(img, sf, min_size, max_size)
    assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
    return p, o, ref


@pytest.mark.parametrize("w,h,sf,seed", [(640, 480, 1.1, 1), (640, 480, 4.0, 2), (333, 127, 1.3, 3), (97, 211, 1.1, 4)])
def test_lbp_windows_bit_exact(lbp_xml, w, h, sf, seed):
    _compare_windows(lbp_xml, frame_natural(w, h, seed), sf, haar=False)


@pytest.mark.parametrize("w,h,sf,seed", [(640, 480, 1.1, 1), (640, 480, 4.0, 2), (333, 127, 1.3, 3), (97, 211, 1.1, 4)])
def test_haar_windows_exact(haar_xml, w, h, sf, seed):
    img = _faces(frame_natural(w, h, seed), seed)
    _compare_windows(haar_xml, img, sf, haar=True)


def test_haar_uniform_noise_and_flat_image(haar_xml):
    _compare_windows(haar_xml, frame_uniform(320, 240, 7), 1.1, haar=True)
    flat = np.full((100, 120), 77, np.uint8)  # zero variance: every window fails the variance test, no skip
    p, o, ref = _compare_windows(haar_xml, flat, 1.1, haar=True)
    assert (ref.codes == -1).all() and len(ref.candidates) == 0
    sat = np.full((400, 400), 255, np.uint8)  # squared-sum integral wraps past 2^32
    sat[::7, ::5] = 0
    _compare_windows(haar_xml, sat, 1.2, haar=True)


def test_min_max_size_filters(haar_xml, lbp_xml):
    img = _faces(frame_natural(480, 360, 21), 21)
    _compare_windows(haar_xml, img, 1.1, True, min_size=(40, 40))
    _compare_windows(haar_xml, img, 1.1, True, max_size=(60, 60))
    _compare_windows(lbp_xml, img, 1.1, False, min_size=(50, 50), max_size=(50, 50))


@pytest.mark.parametrize("which", ["haar", "lbp"])
def test_detect_multiscale_rectangles_identical(haar_xml, lbp_xml, which):
    xml = haar_xml if which == "haar" else lbp_xml
    o = orc.load_cascade_xml(xml)
    p = cc.CascadeClassifier(xml)
    for seed, (w, h) in enumerate([(640, 480), (800, 450), (320, 200)]):
        img = _faces(frame_natural(w, h, 30 + seed), seed)
        for sf, mn in ((1.1, 3), (1.1, 0), (1.25, 2), (4.0, 50), (4.0, 1)):  # (4, 50) = tools/detection/Cpp/main.cpp:45
            a = p.detectMultiScale(img, sf, mn)
            b = orc.detect_multiscale(o, img, sf, mn, nthreads=8)
            assert a.shape == b.shape and (a == b).all(), (which, seed, sf, mn)


def test_tiny_and_degenerate_images(haar_xml):
    p = cc.CascadeClassifier(haar_xml)
    o = orc.load_cascade_xml(haar_xml)
    for (w, h) in [(23, 100), (100, 23), (24, 24), (25, 24), (24, 25), (1, 1)]:
        img = frame_natural(max(w, 2), max(h, 2), 3)[:h, :w]
        a = p.detectMultiScale(img, 1.1, 0)
        b = orc.detect_multiscale(o, img, 1.1, 0)
        assert a.shape == b.shape and (a == b).all()
    with pytest.raises(cc.CascadeError):
        p.detectMultiScale(frame_natural(64, 64, 1), 1.0, 3)  # scaleFactor must be > 1


def test_batch_matches_single_frames(haar_xml):
    frames = np.stack([_faces(frame_natural(400, 300, 50 + i), i) for i in range(5)])
    o = orc.load_cascade_xml(haar_xml)
    for mb in (1, 2, 8):
        p = cc.CascadeClassifier(haar_xml, max_batch=mb)
        got = p.detect_batch(frames, 1.1, 3)
        for i in range(5):
            b = orc.detect_multiscale(o, frames[i], 1.1, 3, nthreads=8)
            assert got[i].shape == b.shape and (got[i] == b).all()


def test_batch_from_device_memory(lbp_xml):
    import torch
    frames = np.stack([frame_natural(320, 240, 70 + i) for i in range(3)])
    t = torch.from_numpy(frames).cuda()
    p = cc.CascadeClassifier(lbp_xml, max_batch=4)
    got = p.detect_batch(None, 1.1, 3, device_ptr=t.data_ptr(), shape=tuple(frames.shape))
    o = orc.load_cascade_xml(lbp_xml)
    for i in range(3):
        b = orc.detect_multiscale(o, frames[i], 1.1, 3, nthreads=8)
        assert got[i].shape == b.shape and (got[i] == b).all()


def test_full_hd_properties(haar_xml):
    """BASELINE size (1920x1080, 40 scales, 4 514 050 windows): full oracle comparison of the candidate set plus
    size-independent properties (window census, visited count, determinism)."""
    img = _faces(frame_natural(1920, 1080, 0), 0, ks=(1.0, 2.0, 5.0, 9.0))
    o = orc.load_cascade_xml(haar_xml)
    p = cc.CascadeClassifier(haar_xml)
    codes, sums, vis = p.debug_windows(img, 1.1)
    assert len(codes) == 4514050
    ref = orc.detect_raw(o, img, 1.1, nthreads=16, full=True)
    assert (codes == ref.codes).all() and (vis == ref.visited).all()
    assert np.max(np.abs(sums - ref.sums)) <= HAAR_SUM_TOL
    assert int(vis.sum()) == ref.n_visited_windows
    raw1 = p.detect_raw(img, 1.1)
    raw2 = p.detect_raw(img, 1.1)
    assert (raw1 == raw2).all() and (raw1 == ref.candidates).all()
    a = p.detectMultiScale(img, 1.1, 3)
    assert (a == orc.detect_multiscale(o, img, 1.1, 3, nthreads=16)).all() and len(a) >= 3


def test_profiling_counters(lbp_xml):
    p = cc.CascadeClassifier(lbp_xml)
    p.set_profiling(True)
    p.detectMultiScale(frame_natural(640, 480, 1), 1.1, 3)
    t = p.timings(reset=True)
    assert t["frames"] == 1 and t["grid_windows"] == 585373 and t["eval_launches"] == 1 and t["eval_ms"] > 0


def test_candidate_list_overflow_in_a_multi_pass_batch(monkeypatch):
    """A weak cascade passes almost every window: the candidate lists overflow on the first pass of a batch while the next
    pass is already in flight with the old capacity. Both must be redone; every frame's rectangles equal the oracle's."""
    from tests import cascade_factory as cf
    cat = orc.haar_catalog(24, 24, 0)
    feats = cat[[1234]].copy()
    xml_text = cf.haar_xml(feats, [(np.float32(-1.0), [([(0, -1, 0, np.float32(0.0))], [1.0, 1.0])])], mode="BASIC")
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".xml", delete=False) as f:
        f.write(xml_text)
        path = f.name
    frames = np.stack([frame_natural(200, 150, 90 + i) for i in range(7)])
    o = orc.load_cascade_xml(path)
    want = [orc.detect_multiscale(o, frames[i], 1.2, 0, nthreads=4) for i in range(7)]
    assert min(len(w) for w in want) > 600  # far more candidates per frame than the initial capacity below
    monkeypatch.setenv("CCAMD_CAND_CAP", "512")
    for mb in (8, 2):
        p = cc.CascadeClassifier(path, max_batch=mb)  # fresh detector: the first call meets the small lists
        for _ in range(2):  # the second call runs with the grown lists
            got = p.detect_batch(frames, 1.2, 0)
            for i in range(7):
                a = got[i][np.lexsort(got[i].T[::-1])]
                b = want[i][np.lexsort(want[i].T[::-1])]
                assert a.shape == b.shape and (a == b).all(), (mb, i)
    os.unlink(path)


def test_single_image_calls_run_from_a_hipgraph(haar_xml, monkeypatch, capfd):
    """The tool's call shape (one host image per call) is launch-bound: after a first ordinary call has sized the buffers
    the pass is captured and replayed as ONE graph launch. cc_detector_graph_active reports it; a capture that fails
    falls back to ordinary launches, says so once, and returns the same rectangles."""
    img = _faces(frame_natural(640, 480, 77), 77)
    want = orc.detect_multiscale(orc.load_cascade_xml(haar_xml), img, 1.1, 3, nthreads=8)
    p = cc.CascadeClassifier(haar_xml)
    a = p.detectMultiScale(img, 1.1, 3)
    assert not p.graph_active()  # first call: ordinary launches, buffers get their sizes
    b = p.detectMultiScale(img, 1.1, 3)
    assert p.graph_active()      # captured and launched as a graph
    c = p.detectMultiScale(img, 1.1, 3)
    assert p.graph_active()      # replayed
    for r in (a, b, c):
        assert r.shape == want.shape and (r == want).all()
    other = _faces(frame_natural(640, 480, 78), 78)
    d = p.detectMultiScale(other, 1.1, 3)  # same geometry, other pixels: still the graph
    assert p.graph_active()
    w2 = orc.detect_multiscale(orc.load_cascade_xml(haar_xml), other, 1.1, 3, nthreads=8)
    assert d.shape == w2.shape and (d == w2).all()
    p.detectMultiScale(img[:300, :400], 1.1, 3)  # new geometry: a new plan starts with an ordinary call
    assert not p.graph_active()
    # forced capture failure: identical rectangles from ordinary launches, one line on stderr
    monkeypatch.setenv("CCAMD_DEBUG_FAIL_CAPTURE", "1")
    q = cc.CascadeClassifier(haar_xml)
    capfd.readouterr()
    for _ in range(3):
        r = q.detectMultiScale(img, 1.1, 3)
        assert not q.graph_active()
        assert r.shape == want.shape and (r == want).all()
    assert capfd.readouterr().err.count("hipGraph capture of the single-image pass failed") == 1


def test_submit_collect_pipelines_batches_with_identical_results(haar_xml, lbp_xml, monkeypatch):
    """cc_detect_batch_submit / cc_detect_batch_collect: batches submitted back to back (the next one before the previous
    one is collected) return exactly what detect_batch returns, whatever else happens in between: another plan (other
    frame size), a single-image call (hipGraph path), a debug call, host and device frames, an overflowing candidate list."""
    import torch
    p = cc.CascadeClassifier(haar_xml, max_batch=8)
    p.specialize(4)
    a = np.stack([frame_natural(480, 270, 100 + i) for i in range(7)])
    b = np.stack([frame_natural(480, 270, 200 + i) for i in range(5)])
    c = np.stack([frame_natural(320, 200, 300 + i) for i in range(3)])  # another plan
    want = {k: p.detect_batch(v, 1.1, 2) for k, v in (("a", a), ("b", b), ("c", c))}
    da = torch.from_numpy(a).cuda()

    def same(got, key):
        assert len(got) == len(want[key]) and all(x.shape == y.shape and (x == y).all() for x, y in zip(got, want[key]))
    t1 = p.detect_batch_submit(a, 1.1, 2)
    t2 = p.detect_batch_submit(b, 1.1, 2)          # fetches t1's last pass while b's first pass runs
    same(p.detect_batch_collect(t1), "a")
    t3 = p.detect_batch_submit(None, 1.1, 2, device_ptr=da.data_ptr(), shape=a.shape)
    same(p.detect_batch_collect(t2), "b")
    t4 = p.detect_batch_submit(c, 1.1, 2)          # other frame size: t3's pending pass is fetched first
    one = p.detectMultiScale(a[0], 1.1, 2)         # single-image call in between (flushes t4's pending pass)
    assert one.shape == want["a"][0].shape and (one == want["a"][0]).all()
    same(p.detect_batch_collect(t4), "c")
    same(p.detect_batch_collect(t3), "a")          # collected out of order
    t5 = p.detect_batch_submit(b, 1.1, 2)
    p.debug_windows(a[1], 1.1)                     # a debug call with a batch pending
    same(p.detect_batch_collect(t5), "b")
    same(p.detect_batch(a, 1.1, 2), "a")           # the synchronous call still works afterwards
    # overflowing candidate lists (tiny initial capacity) while batches overlap
    monkeypatch.setenv("CCAMD_CAND_CAP", "16")
    q = cc.CascadeClassifier(haar_xml, max_batch=8)
    t6 = q.detect_batch_submit(a, 1.1, 2)
    t7 = q.detect_batch_submit(b, 1.1, 2)
    same(q.detect_batch_collect(t6), "a")
    same(q.detect_batch_collect(t7), "b")
    # a batch nobody wants: discarded while it is pending and after another submit fetched it; the detector carries on
    t10 = q.detect_batch_submit(a, 1.1, 2)
    q.detect_batch_discard(t10)
    t11, t12 = q.detect_batch_submit(b, 1.1, 2), q.detect_batch_submit(a, 1.1, 2)
    q.detect_batch_discard(t11)
    q.detect_batch_discard(t11)                    # a second time: nothing left to do
    same(q.detect_batch_collect(t12), "a")
    # LBP (wave phase, 16-bit tiles) through the same path
    r = cc.CascadeClassifier(lbp_xml, max_batch=8)
    r.specialize(20)
    wl = r.detect_batch(a, 1.1, 2)
    t8, t9 = r.detect_batch_submit(a, 1.1, 2), r.detect_batch_submit(a, 1.1, 2)
    for t in (t8, t9):
        got = r.detect_batch_collect(t)
        assert all(x.shape == y.shape and (x == y).all() for x, y in zip(got, wl))


def test_tickets_are_ended_only_by_their_own_detector(haar_xml):
    """Round-3 advisor finding: collect / discard used to free the ticket before checking whose it was, while the owner's
    unfetched pass still delivered into it. Now a ticket handed to the wrong detector stays valid (the call fails), its
    own detector can still collect it, the owner keeps working, and after the owner is gone the ticket can only be
    discarded -- with results checked against the oracle, not against another run of the library."""
    a = np.stack([frame_natural(320, 200, 400 + i) for i in range(5)])
    o = orc.load_cascade_xml(haar_xml)
    want = [orc.detect_multiscale(o, a[i], 1.1, 2, nthreads=4) for i in range(5)]
    p, q = cc.CascadeClassifier(haar_xml, max_batch=8), cc.CascadeClassifier(haar_xml, max_batch=8)
    q.detect_batch(a[:1], 1.1, 2)  # q has a live detector handle too
    t1 = p.detect_batch_submit(a, 1.1, 2)
    out, offs = np.zeros((1024, 4), np.int32), np.zeros(6, np.int32)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    assert L.lib().cc_detect_batch_collect(q._detector(), t1["ticket"], vp(out), 1024, vp(offs)) == L.CC_ERR_INVALID_ARG
    assert L.lib().cc_detect_batch_discard(q._detector(), t1["ticket"]) == L.CC_ERR_INVALID_ARG
    t2 = p.detect_batch_submit(a, 1.1, 2)  # the owner retires t1's pending pass into the (still valid) ticket
    for t in (t1, t2):
        got = p.detect_batch_collect(t)
        for i in range(5):
            x, y = got[i][np.lexsort(got[i].T[::-1])], want[i][np.lexsort(want[i].T[::-1])]
            assert x.shape == y.shape and (x == y).all()
    # the owner is destroyed with a batch outstanding: collecting is an error, discarding (with any detector) frees it
    t3 = p.detect_batch_submit(a, 1.1, 2)
    raw = t3["ticket"]
    p._release()
    assert L.lib().cc_detect_batch_collect(q._detector(), raw, vp(out), 1024, vp(offs)) == L.CC_ERR_INVALID_ARG
    assert L.lib().cc_detect_batch_discard(q._detector(), raw) == L.CC_OK
    got = q.detect_batch(a, 1.1, 2)
    assert all(x.shape == y.shape for x, y in zip(got, want))


def test_host_frames_go_through_the_pinned_staging_area(haar_xml, monkeypatch):
    """Host frames are copied into the detector's pinned staging area and sent with one asynchronous copy per pass
    (round 4): strided frames, a batch that grows the staging areas while a pass of a smaller batch is still unfetched and
    has to be REDONE from its staged frames (candidate-list overflow: the round-3 advisor's second finding), pinned caller
    memory (no staging copy) and the round-3 path (CCAMD_NO_PINNED_STAGING) all give the oracle's rectangles."""
    import torch
    o = orc.load_cascade_xml(haar_xml)
    big = np.stack([frame_natural(360, 240, 500 + i) for i in range(12)])
    want = [orc.detect_multiscale(o, big[i], 1.1, 2, nthreads=4) for i in range(12)]

    def same(got, idx):
        assert len(got) == len(idx)
        for g, i in zip(got, idx):
            x, y = g[np.lexsort(g.T[::-1])], want[i][np.lexsort(want[i].T[::-1])]
            assert x.shape == y.shape and (x == y).all(), i
    monkeypatch.setenv("CCAMD_CAND_CAP", "16")  # every pass overflows its candidate list once and is redone
    p = cc.CascadeClassifier(haar_xml, max_batch=16)
    t1 = p.detect_batch_submit(big[:3], 1.1, 2)      # small staging areas, last pass left pending
    t2 = p.detect_batch_submit(big, 1.1, 2)          # larger passes: the staging areas grow under the pending pass
    same(p.detect_batch_collect(t1), range(3))
    same(p.detect_batch_collect(t2), range(12))
    monkeypatch.delenv("CCAMD_CAND_CAP")
    q = cc.CascadeClassifier(haar_xml, max_batch=16)
    same(q.detect_batch(big, 1.1, 2), range(12))
    pinned = torch.from_numpy(big).pin_memory()
    same(q.detect_batch(pinned.numpy(), 1.1, 2), range(12))  # pinned caller memory: asynchronous copies straight from it
    tk = [q.detect_batch_submit(big[i:i + 4], 1.1, 2) for i in (0, 4, 8)]
    for k, t in enumerate(tk):
        same(q.detect_batch_collect(t), range(4 * k, 4 * k + 4))
    monkeypatch.setenv("CCAMD_NO_PINNED_STAGING", "1")
    r = cc.CascadeClassifier(haar_xml, max_batch=16)
    same(r.detect_batch(big, 1.1, 2), range(12))
