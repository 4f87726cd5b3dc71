"""A scarplet_amd.dist transport between THREADS of one process (launcher-side code, like torch_transport.py): the ranks of
an orientation- or space-sharded search as threads, each with its own context on the same GPU - how the tests run the
multi-rank flows of dist.py against the real device code on the one GPU a test box has.

    shared = ThreadShared(world)
    threads: ThreadTransport(rank, shared)
"""
import threading


class ThreadShared(object):
    def __init__(self, world, timeout=600.0):
        self.world, self.timeout = world, timeout
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.box = None
        self.mail = {}
        self.lock = threading.Lock()


class ThreadTransport(object):
    def __init__(self, rank, shared):
        self.rank, self.s = rank, shared

    def _wait(self):
        self.s.barrier.wait(self.s.timeout)          # (a rank that died breaks the barrier: the others raise instead of hanging)

    def broadcast_bytes(self, payload):
        if self.rank == 0:
            self.s.box = payload
        self._wait()
        out = self.s.box
        self._wait()
        return out

    def gather(self, obj, dst):
        self.s.slots[self.rank] = obj
        self._wait()
        out = list(self.s.slots) if self.rank == dst else None
        self._wait()
        return out

    def exchange(self, sends, recvs):
        import numpy as np
        with self.s.lock:
            for (peer, tag, arr) in sends:
                self.s.mail[(self.rank, peer, tag)] = np.array(arr, dtype=np.float64, copy=True)
        self._wait()
        out = [self.s.mail[(peer, self.rank, tag)].reshape(shape) for (peer, tag, shape) in recvs]
        self._wait()
        with self.s.lock:
            for (peer, tag, _) in recvs:
                self.s.mail.pop((peer, self.rank, tag), None)
        return out


def run_ranks(world, fn, timeout=600.0):
    """fn(rank, transport) on `world` threads; returns their results in rank order, re-raises the first failure."""
    shared = ThreadShared(world, timeout)
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r, ThreadTransport(r, shared))
        except BaseException as e:               # noqa: a failed rank must release the others
            err[r] = e
            shared.barrier.abort()

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    first = [e for e in err if e is not None and not isinstance(e, threading.BrokenBarrierError)] or [e for e in err if e is not None]
    if first:
        raise first[0]
    return out
