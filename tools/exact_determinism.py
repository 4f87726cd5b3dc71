#!/usr/bin/env python3
"""Lab: is exact=True the same from run to run?  The near-tie search's record and event set, then the settled record."""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan
warnings.simplefilter("ignore")
f = np.load(os.path.join(ROOT, "tests/golden/dem_carrizo.npz"))
g = sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"]))
ages, angles = _plan.age_grid(), _plan.angle_grid()
m = sl.Matcher(g)
arr, bbox, area = m.describe(sl.Scarp, 100., ages, angles)
plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
runs = []
for rep in range(4):
    m.ctx.reset_best()
    m.ctx.set_option("near_window", 3.5e-4)
    m.ctx.match(arr, sp, sync=True)
    m.ctx.set_option("near_window", 0.0)
    b0 = [x.copy() for x in m.ctx.get_best()]
    ev = m.ctx.near_events()
    ev = ev[np.lexsort((ev[:, 2], ev[:, 1], ev[:, 0]))]
    fl = m.ctx.near_ties().copy()
    st = m.ctx.settle_exact(len(ages), 0.0)
    b1 = [x.copy() for x in m.ctx.get_best()]
    runs.append((b0, ev, fl, b1, st))
    print(rep, st)
for rep in range(1, 4):
    a, b = runs[0], runs[rep]
    print("run 0 vs", rep, "record before settle equal:", [bool(np.array_equal(x.view(np.uint32), y.view(np.uint32))) for x, y in zip(a[0], b[0])],
          "events equal:", a[1].shape == b[1].shape and bool(np.array_equal(a[1], b[1])), "flags equal:", bool(np.array_equal(a[2], b[2])),
          "record after settle: id differs in", int((a[3][2] != b[3][2]).sum()), "cells, snr in", int((a[3][1] != b[3][1]).sum()))
    d = np.argwhere(a[3][2] != b[3][2])[:6]
    for (i, j) in d:
        print("   cell", i, j, "ids", a[3][2][i, j], b[3][2][i, j], "snr", a[3][1][i, j], b[3][1][i, j], "before", a[0][2][i, j], a[0][1][i, j])
