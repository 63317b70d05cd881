"""Host-side robustness under AddressSanitizer + UBSan (CPU build only; the GPU pool has no sanitizer runs): the cascade
XML reader on thousands of mutated files, the .vec reader on truncated / corrupted files, grouping and the scale plan on
hostile arguments. tests/cpp/fuzz_host.cpp is compiled with g++ against the product's host sources."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import cascade_factory as cf
from tests.util import frame_natural

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cascadeclassifier_amd", "csrc")


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("fuzz") / "fuzz_host")
    cmd = ["g++", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, os.path.join(ROOT, "tests", "cpp", "fuzz_host.cpp")] + [os.path.join(CSRC, f) for f in ("cc_xml.cpp", "cc_cascade.cpp", "cc_host.cpp")] + \
          ["-o", out, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize("which", ["lbp", "haar_trees", "lbp_trees"])
def test_mutated_inputs_never_trip_a_sanitizer(fuzz_bin, which, lbp_xml, repo_root, tmp_path):
    if which == "lbp":
        xml = lbp_xml
    else:
        img = frame_natural(320, 240, 3)
        cal = np.stack([img[y:y + 24, x:x + 24] for y in range(0, 200, 9) for x in range(0, 280, 11)])
        xml = str(tmp_path / "c.xml")
        open(xml, "w").write(cf.haar_tree_cascade(cal, with_tilted=True) if which == "haar_trees" else cf.lbp_tree_cascade())
    r = subprocess.run([fuzz_bin, xml, os.path.join(repo_root, "tests", "golden", "barcode.vec"), "1500", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "loaded" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
