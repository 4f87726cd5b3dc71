"""compare(): running-best fold of host-side results, on the GPU."""

import ctypes as C

import numpy as np


def compare(results, ny, nx, device=0):
    """Reference semantics (core.py:198-243): zeros start state, then for
    every result ``best = (best_snr > snr)*best + (best_snr < snr)*this`` for
    amp, age, angle and (last) snr.  ``age`` and ``angle`` of a result may be
    scalars (match_template's return) or (ny, nx) planes (the (4, ny, nx)
    arrays of calculate_best_fit_parameters, as match() folds them,
    core.py:288-292); both are multiplied cell by cell like the reference's
    numexpr expressions."""
    from scarplet_amd.core import _context
    ctx = _context(device)
    lib, h = ctx.lib, ctx._h
    dp = C.POINTER(C.c_double)

    def plane(a):
        return np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (ny, nx)))

    ctx._check(lib.sc_compare_begin(h, int(ny), int(nx)), "sc_compare_begin")
    for r in results:
        amp, age, angle, snr = r
        amp, snr = plane(amp), plane(snr)
        if np.ndim(age) == 0 and np.ndim(angle) == 0:
            ctx._check(lib.sc_compare_fold(h, amp.ctypes.data_as(dp),
                                           snr.ctypes.data_as(dp), float(age),
                                           float(angle)), "sc_compare_fold")
        else:
            age, angle = plane(age), plane(angle)
            ctx._check(lib.sc_compare_fold_planes(
                h, amp.ctypes.data_as(dp), age.ctypes.data_as(dp),
                angle.ctypes.data_as(dp), snr.ctypes.data_as(dp)),
                "sc_compare_fold_planes")
    out = [np.empty((ny, nx)) for _ in range(4)]
    ctx._check(lib.sc_compare_end(h, *[o.ctypes.data_as(dp) for o in out]),
               "sc_compare_end")
    return tuple(out)
