"""Randomised differential test of the detector against the CPU oracle: image sizes (incl. ones barely larger than the
window and ragged widths), scale factors, min/max object sizes and minNeighbors drawn from fixed seeds; Haar (table-driven
and specialised kernel) and LBP. Everything compared exactly: ungrouped candidates and grouped rectangles."""
import numpy as np
import pytest

import cascadeclassifier_amd as cc
from oracle import oracle as orc
from tests.util import frame_natural, frame_uniform

pytestmark = pytest.mark.gpu


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    for i in range(n):
        w = int(rng.choice([24, 25, 31, 64, 65, 97, 128, 191, 257, 320, 403]))
        h = int(rng.choice([24, 26, 33, 48, 77, 100, 129, 200, 301]))
        sf = float(rng.choice([1.05, 1.1, 1.2, 1.5, 2.0, 3.0, 4.0]))
        mn = int(rng.choice([0, 1, 2, 3, 5]))
        mins = None if rng.random() < 0.6 else (int(rng.integers(24, 60)),) * 2
        maxs = None if rng.random() < 0.6 else (int(rng.integers(40, 200)),) * 2
        content = frame_natural if rng.random() < 0.7 else frame_uniform
        yield w, h, sf, mn, mins, maxs, content(w, h, 1000 * seed + i)


@pytest.mark.parametrize("which", ["haar", "haar_specialised", "lbp"])
def test_random_configurations(which, haar_xml, lbp_xml):
    xml = lbp_xml if which == "lbp" else haar_xml
    o = orc.load_cascade_xml(xml)
    p = cc.CascadeClassifier(xml)
    if which == "haar_specialised":
        assert p.specialize(3) == 3
    total = 0
    for w, h, sf, mn, mins, maxs, img in _cases({"haar": 1, "haar_specialised": 2, "lbp": 3}[which], 40):
        kw, okw = {}, {}
        if mins:
            kw["minSize"] = mins
            okw["min_size"] = mins
        if maxs:
            kw["maxSize"] = maxs
            okw["max_size"] = maxs
        ref = orc.detect_raw(o, img, sf, nthreads=4, **okw)
        raw = p.detect_raw(img, sf, **kw)
        assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all(), (w, h, sf, mins, maxs)
        a = p.detectMultiScale(img, sf, mn, **kw)
        b = orc.detect_multiscale(o, img, sf, mn, nthreads=4, **okw)
        assert a.shape == b.shape and (a == b).all(), (w, h, sf, mn, mins, maxs)
        total += len(raw)
    assert total > 0


def test_large_frames_4k_and_extreme_aspect(haar_xml):
    """Index arithmetic at sizes well beyond Full HD: one 3840x2160 frame and a 8000x40 strip, candidates and grouped
    rectangles identical to the oracle (table-driven and specialised kernel)."""
    o = orc.load_cascade_xml(haar_xml)
    p = cc.CascadeClassifier(haar_xml)
    for img, sf in ((frame_natural(3840, 2160, 77), 1.25), (frame_natural(8000, 40, 78), 1.1)):
        ref = orc.detect_raw(o, img, sf, nthreads=8)
        want = orc.detect_multiscale(o, img, sf, 3, nthreads=8)
        for k in (0, 4):
            p.specialize(k)
            raw = p.detect_raw(img, sf)
            assert raw.shape == ref.candidates.shape and (raw == ref.candidates).all()
            got = p.detectMultiScale(img, sf, 3)
            assert got.shape == want.shape and (got == want).all()


def test_independent_detectors_on_concurrent_host_threads(haar_xml, lbp_xml):
    """Serving shape: several host threads, each with its own detector (own streams, buffers, specialised module), on one
    GPU at the same time. ctypes drops the GIL inside the library, so the calls really overlap."""
    import threading
    frames = np.stack([frame_natural(480, 270, 300 + i) for i in range(6)])
    cfg = [(haar_xml, 0), (haar_xml, 3), (lbp_xml, 0), (haar_xml, 3)]
    serial = []
    for xml, k in cfg:
        p = cc.CascadeClassifier(xml)
        serial.append([r.copy() for r in p.detect_batch(frames, 1.1, 2)])
    results = [None] * len(cfg)
    errors = []

    def work(i):
        try:
            xml, k = cfg[i]
            p = cc.CascadeClassifier(xml)
            if k:
                p.specialize(k)
            out = None
            for _ in range(5):
                out = p.detect_batch(frames, 1.1, 2)
                one = p.detectMultiScale(frames[2], 1.1, 2)
                assert one.shape == out[2].shape and (one == out[2]).all()
            results[i] = out
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(cfg))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for got, want in zip(results, serial):
        assert all(a.shape == b.shape and (a == b).all() for a, b in zip(got, want))


def test_trainer_side_calls_while_another_thread_captures_graphs(haar_xml):
    """A detector's single-image path captures a hipGraph per scale plan. While ANY thread captures, the runtime refuses
    copies on the legacy stream (plain hipMemcpy) in every other thread and fails the capture with them -- round 4 met that
    between two detectors (the specialised modules' tile lists). One thread keeps capturing (new image sizes: a plan and a
    capture each), the other goes through the evaluator's and the detector's entry points that copy: none may fail, and the
    values stay the oracle's."""
    import threading
    from cascadeclassifier_amd import evaluator as ev
    errors, stop = [], threading.Event()

    def capturer():
        try:
            p = cc.CascadeClassifier(haar_xml)
            p.specialize(3)
            k = 0
            while not stop.is_set() and k < 60:
                img = frame_natural(200 + 8 * (k % 30), 150 + 4 * (k % 30), 500 + k)
                a = p.detectMultiScale(img, 1.2, 2)  # sizes the buffers
                b = p.detectMultiScale(img, 1.2, 2)  # captured
                c = p.detectMultiScale(img, 1.2, 2)  # replayed
                assert a.shape == b.shape == c.shape and (a == b).all() and (a == c).all()
                k += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("capturer", repr(e)))

    def trainer():
        try:
            rng = np.random.default_rng(3)
            imgs = rng.integers(0, 256, (40, 24, 24), dtype=np.uint8)
            s, t, nf = orc.set_images(imgs, want_tilted=False, want_norm=True)
            cat = orc.haar_catalog(24, 24, ev.BASIC)
            want = orc.haar_eval_batch(cat, 100, 140, s, t, nf, 24, 24, None)
            for it in range(12):
                e = cc.CvFeatureEvaluator.create(ev.HAAR)
                e.init(cc.CvFeatureParams(ev.HAAR, ev.BASIC), 40, (24, 24))
                e.setImages(imgs, np.zeros(40, np.uint8))
                got = e.calc_batch(100, 140)
                assert (got == want).all()
                assert e(100, 7) == want[0, 7]
                p = cc.CascadeClassifier(haar_xml)
                p.specialize(4)  # tile lists of the specialised modules: a copy per new plan
                p.detect_batch(np.stack([frame_natural(300 + 4 * it, 200, 40 + it)] * 2), 1.2, 2)
                p.debug_windows(frame_natural(120, 90 + it, 9), 1.2)
        except Exception as e:  # noqa: BLE001
            errors.append(("trainer", repr(e)))

    a, b = threading.Thread(target=capturer), threading.Thread(target=trainer)
    a.start()
    b.start()
    b.join()
    stop.set()
    a.join()
    assert not errors, errors
