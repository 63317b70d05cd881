"""Runs the C++ parity test of the host adaptor (tests/cpp/test_features_parity.cpp): the reference's own
test_features.cpp hot-path cases through CvHaarEvaluator / CvLBPEvaluator / Feature::calc, and the detection tool's
call shape, all on the HIP path. CPU part: the adaptor library and test binary exist and link."""
import os
import subprocess

import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cascadeclassifier_amd", "lib")


def test_cpp_adaptor_is_built():
    assert os.path.exists(os.path.join(LIB, "libccamd_cpp.so")), "run __graft_entry__.build()"
    assert os.path.exists(os.path.join(LIB, "test_features_parity"))
    out = subprocess.run(["ldd", os.path.join(LIB, "test_features_parity")], capture_output=True, text=True).stdout
    assert "libccamd_cpp.so" in out and "libcascadeclassifier_amd.so" in out and "not found" not in out


def test_cpp_params_machinery_matches_the_reference_driver_flow():
    """CvParams / CvFeatureParams / CvHaarFeatureParams / CvLBPFeatureParams: create -> printDefaults -> scanAttr -> printAttrs
    -> write -> read, the way traincascade.cpp:59-149 and cascadeclassifier.cpp:200,359-401 drive them (tests/cpp/test_params.cpp;
    no device needed)."""
    r = subprocess.run([os.path.join(LIB, "test_params")], capture_output=True, text=True, timeout=120)
    print(r.stdout[-2000:], r.stderr[-1000:])
    assert r.returncode == 0 and " 0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_feature_tests_pass_on_the_device(haar_xml):
    r = subprocess.run([os.path.join(LIB, "test_features_parity"), haar_xml], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "0 failed" in r.stdout


@pytest.mark.gpu
def test_cpp_detection_tool_matches_python_path(haar_xml, tmp_path):
    """examples/detect_pgm.cpp (the reference's tools/detection/Cpp/main.cpp minus the GUI) prints the same rectangles as
    the Python front end for the same frame."""
    import numpy as np

    import cascadeclassifier_amd as cc
    from tools.make_golden import golden_frame
    img = golden_frame()
    pgm = os.path.join(str(tmp_path), "frame.pgm")
    with open(pgm, "wb") as f:
        f.write(b"P5\n# golden frame\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())
    want = cc.CascadeClassifier(haar_xml).detectMultiScale(img, 1.1, 3)
    r = subprocess.run([os.path.join(LIB, "detect_pgm"), haar_xml, pgm, "1.1", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(v) for v in line.split()] for line in r.stdout.strip().splitlines()], np.int32).reshape(-1, 4)
    assert got.shape == want.shape and (got == want).all() and len(got) >= 3
    assert subprocess.run([os.path.join(LIB, "detect_pgm"), "/nonexistent.xml", pgm], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_unedited_trainer_loop_matches_batched_mining(haar_xml, tmp_path):
    """examples/bench_unedited_trainer.cpp: the reference's window-by-window negative-mining loop (setImage + one
    operator() per weak classifier) through the adaptor's queued setImage / learned-list cache gives every window the
    verdict of cc_negminer_run on the same stream (10 trained stages of the synthetic cascade, 640x480 background)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    trunc = os.path.join(str(tmp_path), "trunc10.xml")
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "truncate_cascade.py"), haar_xml, "10", trunc])
    r = subprocess.run([os.path.join(LIB, "bench_unedited_trainer"), trunc], capture_output=True, text=True, timeout=900)
    print(r.stdout[-2000:], r.stderr[-1000:])
    assert r.returncode == 0
    assert "verdicts identical to the batched path: yes" in r.stdout
