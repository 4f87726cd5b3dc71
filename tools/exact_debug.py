#!/usr/bin/env python3
"""Lab: the cells of a window where exact=True is off the oracle - what the record, the events and the oracle say."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import multiprocessing as mp
import scarplet_oracle as orc
pool = mp.get_context("fork").Pool(48)
import scarplet_amd as sl
from scarplet_amd import _plan
warnings.simplefilter("ignore")
f = np.load(os.path.join(ROOT, "tests/golden/dem_carrizo.npz"))
z, dx, dy = f["z"].astype(float), float(f["dx"]), float(f["dy"])
g = sl.DEMGrid.from_array(z, dx, dy)
ages, angles = _plan.age_grid(), _plan.angle_grid()
T = len(ages) * len(angles)
par, ang = np.repeat(ages, len(angles)), np.tile(angles, len(ages))
win = (420, 452, 230, 263)
i0, i1, j0, j1 = win
a_st, s_st = orc.snr_stack_window(z, dx, dy, orc.SCARP, 100., ages, angles, win, 90, pool=pool)
a_st, s_st = a_st.reshape(T, i1 - i0, j1 - j0), s_st.reshape(T, i1 - i0, j1 - j0)
P = orc.PARITY
for method in ("fft", "auto"):
  for exact in (False, True):
    m = sl.Matcher(g)
    m.search(sl.Scarp, 100., ages, angles, method=method, exact=exact)
    res = m.result()
    sub = tuple(np.asarray(r)[i0:i1, j0:j1] for r in res)
    chk = orc.check_fold(sub, a_st, s_st, par, ang, tie_rtol=orc.tie_window("fft", orc.SCARP),
                         amp_tol=(P["amp"][0], P["amp"][1] * np.max(np.abs(a_st))), snr_tol=(P["snr"][0], P["snr"][1] * np.max(s_st)))
    print(method, "exact", exact, m.method_used, "bad", chk["n_bad"], "inexact", chk["n_inexact"], "snr_err %.2e" % chk["snr_err"], getattr(m, "exact_stats", None) if exact else "")
    if exact:
        flags = m.ctx.near_ties()
    bad = np.argwhere(~chk["ok"])
    for (a, b) in bad[:12]:
        s = s_st[:, a, b]
        t_arg = int(np.argmax(s))
        cand = np.flatnonzero(s >= s[t_arg] * (1 - 1e-3))
        rr = [float(r[a, b]) for r in sub]
        tt = np.flatnonzero((par == rr[1]) & (ang == rr[2]))
        print("  cell", i0 + a, j0 + b, "result amp %.6g age %.4g ang %+.4f snr %.8g" % tuple(rr), "| flagged" if exact and flags[i0 + a, j0 + b] else "",
              "| oracle argmax t=%d (%.4g, %+.4f) snr %.8g amp %.6g" % (t_arg, par[t_arg], ang[t_arg], s[t_arg], a_st[t_arg, a, b]),
              "| oracle at the result's template: snr %.8g amp %.6g" % ((s[tt[0]], a_st[tt[0], a, b]) if len(tt) else (np.nan, np.nan)),
              "| within 1e-3:", [(int(c), "%.6g" % s[c]) for c in cand[:8]])
pool.terminate()
