"""scarplet_amd._hostpool: result blocks are recycled only when no view of them is alive."""
import numpy as np

from scarplet_amd import _hostpool as hp


def test_blocks_are_reused_only_when_unreferenced(monkeypatch):
    monkeypatch.setattr(hp, "MIN_BYTES", 1024)
    monkeypatch.setattr(hp, "_blocks", [])
    monkeypatch.setattr(hp, "MAX_BLOCKS", 2)
    a = hp.empty((4, 16, 16))
    assert a.dtype == np.float64 and a.shape == (4, 16, 16) and a.flags.c_contiguous
    pa = a.ctypes.data
    v = a[1]                                   # a slice the caller keeps
    w = v.reshape(-1)[3:9]                     # ... and a view of a view
    del a, v
    b = hp.empty((4, 16, 16))
    assert b.ctypes.data != pa                 # still referenced through w: not handed out again
    del w
    c = hp.empty((4, 16, 16))
    assert c.ctypes.data == pa                 # every view dropped: recycled
    d = hp.empty((4, 16, 16))                  # b and c alive: a third block, not kept beyond MAX_BLOCKS
    assert d.ctypes.data not in (b.ctypes.data, c.ctypes.data) and len(hp._blocks) == hp.MAX_BLOCKS
    e = hp.empty((2, 8))                       # small: plain numpy
    assert e.base is None
    del b, c, d
    hp.release()
    assert hp._blocks == []


def test_one_block_is_pinned_by_default_and_threads_never_share_one(monkeypatch):
    import threading
    assert hp.MAX_BLOCKS == 1                  # (SCARPLET_HOSTPOOL_BLOCKS unset: 3.2 GB at 10000 x 10000, not 6.4)
    monkeypatch.setattr(hp, "MIN_BYTES", 1024)
    monkeypatch.setattr(hp, "_blocks", [])
    got, go = [], threading.Barrier(8)

    def worker():
        go.wait()
        for _ in range(50):
            got.append(hp.empty((4, 16, 16)))

    th = [threading.Thread(target=worker) for _ in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert len({a.ctypes.data for a in got}) == len(got)      # all alive: all distinct
    assert len(hp._blocks) <= hp.MAX_BLOCKS
    got.clear()
    hp.release()
    assert hp._blocks == []
    monkeypatch.setattr(hp, "MAX_BLOCKS", 0)                  # the opt-out: plain numpy
    assert hp.empty((4, 16, 16)).base is None


def test_prefault_hands_the_touched_block_to_the_next_request(monkeypatch):
    """Matcher.search faults the result block in while the device works: the block the prefault thread touched is
    the one the next empty() of that size returns; a second prefault with that block waiting does nothing."""
    monkeypatch.setattr(hp, "MIN_BYTES", 1024)
    monkeypatch.setattr(hp, "_blocks", [])
    monkeypatch.setattr(hp, "MAX_BLOCKS", 1)
    t = hp.prefault((4, 32, 32))
    assert t is not None
    t.join()
    assert len(hp._blocks) == 1 and hp._free_at(0) and not hp._blocks[0].any()
    assert hp.prefault((4, 32, 32)) is None                  # one is waiting
    a = hp.empty((4, 32, 32))
    assert a.ctypes.data == hp._blocks[0].ctypes.data
    assert hp.prefault((2, 4)) is None                       # small results: plain numpy
    del a
    hp.release()
