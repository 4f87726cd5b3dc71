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
"""
import sys

import numpy as np

MIN_BYTES = 64 << 20          # smaller results: plain np.empty
MAX_BLOCKS = 2                # blocks kept; beyond it unreferenced blocks are dropped first
_blocks = []


def _free_at(i):
    # references to a block nobody outside holds: the _blocks list and getrefcount's own argument
    return sys.getrefcount(_blocks[i]) == 2


def empty(shape, dtype=np.float64):
    """Like ``np.empty(shape, dtype)``; large arrays are views of recycled blocks."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    if n < MIN_BYTES:
        return np.empty(shape, dtype=dtype)
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
    for i in reversed(range(len(_blocks))):
        if _free_at(i):
            del _blocks[i]
