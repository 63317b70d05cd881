import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # keep the specialised-kernel cache of this test session out of the user's home
    if "CCAMD_CACHE_DIR" not in os.environ:
        import tempfile
        os.environ["CCAMD_CACHE_DIR"] = tempfile.mkdtemp(prefix="ccamd_cache_")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def lbp_xml():
    return os.path.join(ROOT, "data", "lbpcascade_frontalface.xml")


@pytest.fixture(scope="session")
def haar_xml():
    return os.path.join(ROOT, "data", "haarcascade_frontalface_synthetic.xml")
