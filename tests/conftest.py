import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; compiled on demand with gcc)."""
    from oracle import oracle as orc

    orc.build()
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np

    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "golden_small.json")) as f:
        small = json.load(f)
    kat = np.load(os.path.join(g, "pack_kat.npz"), allow_pickle=False)
    return small, kat
