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


def _gpu_box():
    """A GPU box, judged WITHOUT a HIP call: collection must not initialise the
    GPU in this process (tests fork pools and start subprocesses later, and a
    process that has initialised the GPU must not fork or exec on the GPU boxes)."""
    lib = os.path.join(ROOT, "scarplet_amd", "libscarplet_hip.so")
    return os.path.exists("/dev/kfd") and os.path.exists(lib)


def pytest_collection_modifyitems(config, items):
    """Without a device (or without the built library) the gpu-marked tests
    are skipped rather than failed; `-m "not gpu"` deselects them anyway."""
    if _gpu_box():
        return
    skip = pytest.mark.skip(reason="needs an MI355X and the built libscarplet_hip.so")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def oracle_pool(request):
    """Process pool for the oracle stacks of the gpu tests.  Created at session
    start, before any test can create a GPU context: the workers are forked from
    a process that has not touched HIP and only ever run numpy."""
    want = _gpu_box() and any("gpu" in item.keywords for item in request.session.items)
    if not want:
        yield None
        return
    import multiprocessing as mp
    import scarplet_oracle  # noqa: F401  (workers inherit the module)
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    pool = mp.get_context("fork").Pool(max(1, min(n, 96)))
    yield pool
    pool.terminate()
    pool.join()


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
def gpu_ctx(oracle_pool):
    from scarplet_amd import _lib
    return _lib.Context(0)
