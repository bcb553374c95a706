import glob
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_analysis_names():
    return sorted(os.path.basename(p)[len("analysis_"):-4] for p in glob.glob(os.path.join(GOLDEN, "analysis_*.npz")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def hip():
    """The loaded C-ABI library wrapper; GPU tests fail loudly if it is missing."""
    from aurora_amd import _lib
    return _lib.load()
