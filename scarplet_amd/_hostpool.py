"""Recycled host buffers for the (4, ny, nx) float64 results.

``sl.match`` returns 32 bytes per cell - 3.2 GB for a 10000 x 10000 DEM.  A fresh
``np.empty`` of that size is fresh pages from the kernel: faulting them in under the
device-to-host copy costs 24 ms per GB (measured here: 24 GB/s into untouched memory
against 56 GB/s into memory that has been written before), 77 ms of every call.  So large
results are views of owner blocks that this module keeps: a block whose last view the
caller has dropped is handed out again, already faulted in.

Safety: a block is reused only when NOTHING outside this module references it.  Every
numpy view of a block - the returned array, any slice or reshape of it - holds a
reference to the owner block itself (numpy collapses view chains onto the owner), so
``sys.getrefcount`` of the block tells whether the caller still holds any of them.
The check and the view that follows it run under one lock: two threads asking for a
result block at once (one Matcher per device) are never handed the same block.

Memory: the module pins at most MAX_BLOCKS blocks (one by default: a result the caller has
dropped stays faulted in for the next call of the same size - 3.2 GB for a 10000 x 10000
DEM; a caller that still HOLDS the previous result gets a fresh, unrecycled block, as
without the pool: drop results before the next call, or set SCARPLET_HOSTPOOL_BLOCKS=2).  ``scarplet_amd.release_host_buffers()`` gives unreferenced blocks back to the
system, and ``SCARPLET_HOSTPOOL_BLOCKS=0`` in the environment switches the recycling off.
"""
import os
import sys
import threading

import numpy as np

MIN_BYTES = 64 << 20          # smaller results: plain np.empty
# blocks kept; beyond it unreferenced blocks are dropped first
MAX_BLOCKS = max(0, int(os.environ.get("SCARPLET_HOSTPOOL_BLOCKS", "1")))
_blocks = []
_lock = threading.Lock()


def _free_at(i):
    # references to a block nobody outside holds: the _blocks list and getrefcount's own argument
    return sys.getrefcount(_blocks[i]) == 2


def empty(shape, dtype=np.float64):
    """Like ``np.empty(shape, dtype)``; large arrays are views of recycled blocks."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    if n < MIN_BYTES or MAX_BLOCKS == 0:
        return np.empty(shape, dtype=dtype)
    with _lock:
        for i in range(len(_blocks)):
            if _blocks[i].nbytes == n and _free_at(i):
                return _blocks[i].view(dtype).reshape(shape)
        # none to reuse: drop blocks nobody holds (of other sizes), keep at most MAX_BLOCKS
        for i in reversed(range(len(_blocks))):
            if len(_blocks) >= MAX_BLOCKS and _free_at(i):
                del _blocks[i]
        blk = np.empty(n, dtype=np.uint8)
        if len(_blocks) < MAX_BLOCKS:
            _blocks.append(blk)
        return blk.view(dtype).reshape(shape)


def release():
    """Drop every block no caller references (the memory goes back to the system)."""
    with _lock:
        for i in reversed(range(len(_blocks))):
            if _free_at(i):
                del _blocks[i]


def prefault(shape, dtype=np.float64):
    """Make sure a block for ``empty(shape, dtype)`` exists and its pages are faulted in - from a thread of its own,
    while the caller waits for the device.  The (4, ny, nx) float64 result of a 10000 x 10000 search is 3.2 GB: into
    fresh pages the device-to-host copy runs at 24 GB/s instead of 56 (75 ms of a first call); touched beforehand -
    the search itself takes seconds - the first call of a process costs what a repeated one does.  Returns the
    thread (already started), or None when there is nothing to do; the block is kept by the pool like any other
    and handed out by the next ``empty`` of that size."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    if n < MIN_BYTES or MAX_BLOCKS == 0:
        return None
    with _lock:
        for i in range(len(_blocks)):
            if _blocks[i].nbytes == n and _free_at(i):
                return None                              # a recycled block of that size is waiting already
        # (advisor, round 5) a block the pool cannot KEEP is not worth touching: with every slot held by a caller - the
        # usual `r = match(...)` loop holding the previous result - empty() hands out an unregistered block, the thread
        # would zero 3.2 GB that are freed when it ends, and get_result would take yet another fresh block
        held = sum(0 if _free_at(i) else 1 for i in range(len(_blocks)))
        if held >= MAX_BLOCKS:
            return None
    blk = empty(shape, dtype)                            # (registers a fresh block with the pool: there is room)

    def touch(a):
        flat = a.reshape(-1).view(np.uint8)
        step = 1 << 26                                   # numpy releases the GIL inside these fills
        for o in range(0, flat.size, step):
            flat[o:o + step] = 0

    t = threading.Thread(target=touch, args=(blk,), daemon=True)
    del blk                                              # (the thread holds the only view; when it ends the block is free)
    t.start()
    return t
