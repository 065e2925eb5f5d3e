import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no built artefacts (they are git-ignored): build them
    # once so the suite can run.  This is test set-up only -- the product never
    # builds or falls back at run time (a missing libd2pc.so is an ImportError).
    lib = os.path.join(ROOT, "disparity_to_point_cloud_amd", "libd2pc.so")
    ora = os.path.join(ROOT, "oracle", "libd2pc_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
