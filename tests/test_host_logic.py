"""CPU-side checks of the product library: it loads, exports every symbol include/cascadeclassifier_amd.h declares,
parses cascade XML like the independent (ElementTree) reader of the oracle, reproduces the oracle's pyramid geometry
and rectangle grouping, and fails LOUDLY when there is no HIP device. No compute entry point is exercised here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cascadeclassifier_amd as cc
from cascadeclassifier_amd import _lib as L
from oracle import oracle as orc


def _declared_symbols(repo_root):
    text = open(os.path.join(repo_root, "include", "cascadeclassifier_amd.h")).read()
    return sorted(set(re.findall(r"CC_API\s+[\w\s\*]+?\b(cc_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(repo_root):
    names = _declared_symbols(repo_root)
    assert len(names) >= 35
    raw = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in the header but not exported"
    assert sorted(L.SIGNATURES) == names, "ctypes binding table and header disagree"


def test_version_and_error_channel():
    assert L.lib().cc_version() >= 100
    st = L.lib().cc_cascade_load_xml(b"/nonexistent/cascade.xml", C.byref(C.c_void_p()))
    assert st == L.CC_ERR_IO
    assert b"cannot open" in L.lib().cc_last_error()


@pytest.mark.parametrize("name", ["lbpcascade_frontalface.xml", "haarcascade_frontalface_synthetic.xml"])
def test_xml_reader_matches_independent_parse(repo_root, name):
    path = os.path.join(repo_root, "data", name)
    o = orc.load_cascade_xml(path)
    p = cc.CascadeClassifier(path)
    assert not p.empty()
    m = p.model()
    assert m.info["feature_type"] == o.feature_type
    assert (m.info["win_w"], m.info["win_h"]) == (o.win_w, o.win_h)
    assert m.info["max_nodes_per_tree"] == 1
    assert (m.stage_ntrees == o.stage_ntrees).all()
    # product stores (float)thr - 1e-5f (what the detector compares against); the oracle applies the eps itself
    assert (m.stage_threshold == (o.stage_threshold - np.float32(1e-5)).astype(np.float32)).all()
    assert (m.stump_feature == o.stump_feature).all()
    assert (m.stump_left == o.stump_left).all() and (m.stump_right == o.stump_right).all()
    if o.feature_type == 0:
        assert (m.stump_threshold == o.stump_threshold).all()
        assert (m.rects == o.haar["r"]).all() and (m.weights == o.haar["wt"]).all()
        assert (m.tilted == o.haar["tilted"]).all()
    else:
        assert (m.stump_subsets.reshape(-1) == o.stump_subset).all()
        assert (m.rects == o.lbp_rect).all()


def test_stock_lbp_cascade_shape(lbp_xml):
    p = cc.CascadeClassifier(lbp_xml)
    inf = p.info()
    assert (inf["n_stages"], inf["n_weak"], inf["n_features"], inf["subset_size"]) == (20, 139, 136, 8)


MINI = """<?xml version="1.0"?><opencv_storage><cascade><stageType>BOOST</stageType><featureType>HAAR</featureType>
<height>24</height><width>24</width><featureParams><maxCatCount>0</maxCatCount></featureParams><stageNum>1</stageNum>
<stages><_><maxWeakCount>1</maxWeakCount><stageThreshold>-1.5</stageThreshold><weakClassifiers>
<_><internalNodes>0 -1 0 4.5e-03</internalNodes><leafValues>-1. 1.</leafValues></_></weakClassifiers></_></stages>
<features><_><rects><_>%s</_><_>7 4 3 9 2.</_></rects><tilted>0</tilted></_></features></cascade></opencv_storage>"""


def test_xml_reader_rejects_bad_input():
    p = cc.CascadeClassifier()
    assert p.load_from_string(MINI % "4 4 9 9 -1.")
    assert p.info()["n_features"] == 1
    assert not p.load_from_string(MINI % "20 4 9 9 -1.")  # rect leaves the window: would index outside the integral
    assert "leaves the 24x24 window" in p.load_error
    assert not p.load_from_string("<opencv_storage><cascade><stageType>BOOST</stageType>")
    assert not p.load_from_string(MINI.replace("BOOST", "SVM") % "4 4 9 9 -1.")
    assert not p.load_from_string(MINI.replace("0 -1 0 4.5e-03", "0 -1 7 4.5e-03") % "4 4 9 9 -1.")  # featureIdx out of range
    assert p.empty()
    # old-format cascades are refused with a message, not mis-parsed
    assert not p.load_from_string("<opencv_storage><haarcascade type_id='opencv-haar-classifier'><size>24 24</size>"
                                  "<stages></stages></haarcascade></opencv_storage>")
    assert "old-format" in p.load_error


@pytest.mark.parametrize("w,h,sf,mn,mx", [
    (1920, 1080, 1.1, None, None), (640, 480, 4.0, None, None), (640, 480, 1.1, None, None), (1920, 1080, 4.0, None, None),
    (24, 24, 1.1, None, None), (25, 31, 1.05, None, None), (333, 127, 1.3, None, None), (640, 480, 1.1, (60, 60), None),
    (640, 480, 1.1, None, (100, 100)), (640, 480, 1.1, (50, 50), (50, 50)), (1281, 723, 1.2, (30, 30), (400, 400)),
    (23, 100, 1.1, None, None), (3840, 2160, 1.1, None, None)])
def test_scale_plan_matches_oracle(w, h, sf, mn, mx):
    a = cc.scale_plan(24, 24, w, h, sf, mn, mx)
    b = orc.scales(24, 24, w, h, sf, mn or (0, 0), mx or (0, 0))
    assert len(a) == len(b)
    for f in ("scale", "w", "h", "ystep", "nx", "ny", "win_w", "win_h"):
        assert (a[f] == b[f]).all(), f


def test_scale_plan_survey_numbers():
    # SURVEY.md Appendix C
    for (w, h, sf, ns, nwin) in [(640, 480, 4, 3, 84169), (640, 480, 1.1, 32, 585373), (1920, 1080, 4, 3, 619265),
                                 (1920, 1080, 1.1, 40, 4514050)]:
        sc = cc.scale_plan(24, 24, w, h, sf)
        assert len(sc) == ns
        assert int((sc["nx"].astype(np.int64) * sc["ny"]).sum()) == nwin
    sc = cc.scale_plan(24, 24, 1920, 1080, 1.1)
    assert int(((sc["w"] + 1).astype(np.int64) * (sc["h"] + 1)).sum()) == 11976323


def test_group_rectangles_matches_oracle():
    rng = np.random.default_rng(3)
    for trial in range(40):
        n = int(rng.integers(0, 120))
        centers = rng.integers(0, 600, (max(n // 6, 1), 2))
        rects = []
        for i in range(n):
            c = centers[rng.integers(0, len(centers))]
            s = int(rng.integers(24, 90))
            rects.append([c[0] + rng.integers(-6, 7), c[1] + rng.integers(-6, 7), s, s])
        rects = np.array(rects, np.int32).reshape(-1, 4)
        for thr in (0, 1, 2, 3, 5):
            a = cc.group_rectangles(rects, thr, 0.2)
            b = orc.group_rectangles(rects, thr, 0.2)
            assert a.shape == b.shape and (a == b).all()


def test_group_rectangles_hand_case():
    # three near-identical rects + one loner, minNeighbors 2 -> only the cluster average survives
    rects = np.array([[10, 10, 40, 40], [12, 11, 40, 40], [11, 12, 42, 42], [300, 300, 24, 24]], np.int32)
    out = cc.group_rectangles(rects, 2, 0.2)
    assert out.tolist() == [[11, 11, 41, 41]]
    assert cc.group_rectangles(rects, 0, 0.2).tolist() == rects.tolist()


def test_no_device_is_a_loud_error(lbp_xml):
    if L.lib().cc_device_count() > 0:
        pytest.skip("a HIP device is present")
    p = cc.CascadeClassifier(lbp_xml)
    with pytest.raises(cc.CascadeError) as ei:
        p.detectMultiScale(np.zeros((64, 64), np.uint8))
    assert ei.value.status == L.CC_ERR_NO_DEVICE
    e = cc.CvFeatureEvaluator.create(1)
    with pytest.raises(cc.CascadeError) as ei:
        e.init(cc.CvFeatureParams(1), 4, (24, 24))
    assert ei.value.status == L.CC_ERR_NO_DEVICE


def test_product_never_imports_the_oracle(repo_root):
    pkg = os.path.join(repo_root, "cascadeclassifier_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), f"{f} mentions the oracle"


def test_specialised_kernel_source_compiles_for_gfx950(haar_xml):
    """cc_detector_specialize's generated source (cascade constants as immediates) through hiprtc, compile only: the CPU-side
    build check of the run-time path. Skipped where libhiprtc is not installed."""
    import ctypes as C

    from cascadeclassifier_amd import _lib as L
    lib = L.lib()
    c = C.c_void_p()
    L.check(lib.cc_cascade_load_xml(haar_xml.encode(), C.byref(c)))
    n = C.c_size_t(0)
    st = lib.cc_cascade_compile_specialized(c, 2, b"gfx950", C.byref(n))
    msg = lib.cc_last_error().decode()
    lib.cc_cascade_destroy(c)
    if st == L.CC_ERR_UNSUPPORTED and "libhiprtc" in msg:
        pytest.skip(msg)
    assert st == L.CC_OK, msg
    assert n.value > 10000
    # the code object is now in the per-process cache and on disk ($CCAMD_CACHE_DIR): same size, no recompilation
    import glob
    import os
    import time
    assert glob.glob(os.path.join(os.environ["CCAMD_CACHE_DIR"], "spec_*.hsaco"))
    L.check(lib.cc_cascade_load_xml(haar_xml.encode(), C.byref(c)))
    m = C.c_size_t(0)
    t0 = time.perf_counter()
    L.check(lib.cc_cascade_compile_specialized(c, 2, b"gfx950", C.byref(m)))
    assert m.value == n.value and time.perf_counter() - t0 < 0.5
    lib.cc_cascade_destroy(c)


def test_disk_cache_of_specialised_code_checks_its_header(haar_xml, tmp_path):
    """A cached code object is only loaded if its header repeats the key's length and hashes (the key covers hiprtc
    version, options and source): a file for another key -- simulated by damaging the header -- is ignored, the kernel is
    recompiled and the file rewritten. Each step in its own process (the in-process cache would hide the file)."""
    import glob
    import subprocess
    import sys
    cache = str(tmp_path)
    code = ("import ctypes as C, sys, time; sys.path.insert(0, %r); from cascadeclassifier_amd import _lib as L; lib = L.lib(); c = C.c_void_p(); "
            "L.check(lib.cc_cascade_load_xml(%r.encode(), C.byref(c))); n = C.c_size_t(0); t = time.perf_counter(); "
            "st = lib.cc_cascade_compile_specialized(c, 2, b'gfx950', C.byref(n)); print(st, n.value, time.perf_counter() - t, lib.cc_last_error().decode())"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), haar_xml))
    env = dict(os.environ, CCAMD_CACHE_DIR=cache)

    def run():
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        st, size, secs = r.stdout.split()[:3]
        return int(st), int(size), float(secs), r.stdout

    st, size, t_first, out = run()
    if st != 0 and "libhiprtc" in out:
        pytest.skip(out)
    assert st == 0
    files = sorted(glob.glob(os.path.join(cache, "spec_*.hsaco")))
    assert len(files) == 2  # a Haar kernel with 32-bit tiles is one module per step (round 4): one cached code object each
    blobs = [open(f, "rb").read() for f in files]
    assert all(b[:8] == b"CCAMDSP2" and b[32:36] == b"\x7fELF" for b in blobs) and sum(len(b) - 32 for b in blobs) == size
    st, size2, t_cached, _ = run()
    assert st == 0 and size2 == size and t_cached < t_first * 0.5  # served from disk
    damaged = bytearray(blobs[0])
    damaged[24] ^= 0x5A  # second hash of the key: as if the file belonged to another source
    open(files[0], "wb").write(bytes(damaged))
    st, size3, t_again, _ = run()
    assert st == 0 and size3 == size and t_again > t_cached * 2  # ignored: compiled again ...
    assert open(files[0], "rb").read() == blobs[0]               # ... and written back with a matching header
    open(files[1], "wb").write(blobs[1][:40])  # truncated file: ignored as well
    st, size4, _, _ = run()
    assert st == 0 and size4 == size and open(files[1], "rb").read() == blobs[1]


def test_value_cache_policy_of_the_cpp_adaptor():
    """ccamd/value_cache_policy.hpp (the hit / miss bookkeeping behind CvFeatureEvaluator::operator()) replayed on the
    trainer's two access shapes: precalculate's row walk costs one launch per feature and teaches the learned list
    nothing; a mining walk costs one launch per window once the cascade's features are learned (round-2 advisor finding:
    the first sample of every row used to be taken for a prediction walk). tests/cpp/test_cache_policy.cpp, no GPU."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cascadeclassifier_amd", "lib", "test_cache_policy")
    assert os.path.exists(exe), "build first: make -C cascadeclassifier_amd/cpp (or __graft_entry__.build())"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "test_cache_policy OK" in r.stdout, r.stdout + r.stderr
