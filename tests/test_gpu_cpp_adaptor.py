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


@pytest.mark.gpu
def test_cpp_feature_tests_pass_on_the_device(haar_xml):
    r = subprocess.run([os.path.join(LIB, "test_features_parity"), haar_xml], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "0 failed" in r.stdout
