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


def _gpu_available():
    try:
        from scarplet_amd import _lib
        return _lib.load().sc_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """Without a device (or without the built library) the gpu-marked tests
    are skipped rather than failed; `-m "not gpu"` deselects them anyway."""
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built libscarplet_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
