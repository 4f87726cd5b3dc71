#!/usr/bin/env python3
"""exact=True on the headline workload (BASELINE config C3: 10000 x 10000, 35 ages x 181 orientations): what the mode costs
there and - on a few windows against ALL 6335 templates - that every decidable cell carries the oracle's argmax.
usage: python tools/exact_c3.py [windows]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import multiprocessing as mp
import scarplet_oracle as orc
from scarplet_amd import _plan, synthetic

n = 10000
g = synthetic.synthetic_scarp(n)
ages, angles = _plan.age_grid(), _plan.angle_grid()
pool = mp.get_context("fork").Pool(min(64, len(os.sched_getaffinity(0))))      # before the first HIP call
import scarplet_amd as sl
sl.Matcher.EXACT_MAX_F64 = float(os.environ.get("EXACT_MAX_F64", sl.Matcher.EXACT_MAX_F64))
m = sl.Matcher(g)
for exact in (False, True):
    t0 = time.time()
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        m.search(sl.Scarp, 100, ages, angles, method="fft", exact=exact)
        res = m.result()
    print("exact=%s: search + result %.2f s  %s  %s" % (exact, time.time() - t0, getattr(m, "exact_stats", "") if exact else "",
                                                        [str(w.message)[:160] for w in wl]), flush=True)
p = m.plan
V, w = p.Vy, 48
wins = {"tile interior": (800, 900), "corner of tiles (1,1)": (V - 24, V - 24), "row seam 2|3": (3 * V - 24, 4000),
        "wrap corner": (0, n - w), "partial last tiles": (n - 400, n - 420), "centre of the scarp": (5000 - 24, 5000 - 24)}
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T = len(ages) * len(angles)
P = orc.PARITY
tot_inexact = 0
for name, (i0, j0) in list(wins.items())[:k]:
    win = (i0, i0 + w, j0, j0 + w)
    a_st, s_st = orc.snr_stack_window(g._griddata, 1.0, 1.0, orc.SCARP, 100, ages, angles, win, 160, pool=pool)
    sub = tuple(np.asarray(r)[win[0]:win[1], win[2]:win[3]] for r in res)
    chk = orc.check_fold(sub, a_st.reshape(T, w, w), s_st.reshape(T, w, w), np.repeat(ages, len(angles)),
                         np.tile(angles, len(ages)), tie_rtol=orc.tie_window("fft", orc.SCARP),
                         amp_tol=(P["amp"][0], P["amp"][1] * np.max(np.abs(a_st))), snr_tol=(P["snr"][0], P["snr"][1] * np.max(s_st)))
    print("window %-24s (%5d, %5d)  bad=%d off the argmax=%d  strict=%d tie=%d of %d" % (name, i0, j0, chk["n_bad"], chk["n_inexact"],
                                                                                          chk["n_strict"], chk["n_tie"], chk["n"]), flush=True)
    assert chk["n_bad"] == 0, name
    tot_inexact += chk["n_inexact"]
print("cells off the oracle's argmax with exact=True:", tot_inexact)
pool.terminate()
