#!/usr/bin/env python3
"""Time a RightFacingUpperBreakScarp search (error masks: the FULL row kernel) - tools/upper_break_lab.py [n]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
g = synthetic.synthetic_scarp(n)
m = sl.Matcher(g)
ages, angs = _plan.age_grid(), _plan.angle_grid()[::12]
for rep in range(3):
    m.ctx.profile(4)
    t0 = time.perf_counter()
    m.search(sl.RightFacingUpperBreakScarp, 100, ages, angs, method="fft")
    dt = time.perf_counter() - t0
    print("rep %d: %.3f s  %s  %s" % (rep, dt, m.plan, {k: round(v[1], 1) for k, v in m.ctx.profile_get().items() if v[0]}), flush=True)
best = m.ctx.get_best()
print("checksum", float(best[1].sum()), int(best[2].astype(np.uint64).sum()))
