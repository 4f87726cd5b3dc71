import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def golden(name):
    return os.path.join(GOLDEN, name)


def load_cases(name):
    """Fixture files written by oracle/gen_golden.py store case i's field k
    as '<k>_<i>' plus a count 'n'."""
    f = np.load(golden(name), allow_pickle=False)
    out = []
    for i in range(int(f["n"])):
        d = {}
        for key in f.files:
            if key.endswith("_%d" % i):
                d[key[:-len("_%d" % i)]] = f[key]
        out.append(d)
    return out


@pytest.fixture(scope="session")
def gpu_ctx():
    from scarplet_amd import _lib
    return _lib.Context(0)
