#!/usr/bin/env python3
"""The settle's cost on a small search (BASELINE config C5, one scale at a time): sc_match with the window on, then
sc_settle_exact alone, timed over repeats; option "variant" 20 = one wave per pair whatever the list's length.
usage: python tools/settle_small.py [variant]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan

f = np.load(os.path.join(ROOT, "tests/golden/dem_grandcanyon.npz"))
g = sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"]))
m = sl.Matcher(g)
if len(sys.argv) > 1:
    m.ctx.set_option("variant", float(sys.argv[1]))
angles = _plan.angle_grid()
for scale in (5.0, 10.0, 20.0, 40.0, 80.0):
    arr, bbox, area = m.describe(sl.Channel, scale, np.array([0.1]), angles)
    m.plan, sp = m.plan_for(bbox, area, "fft", None, n_params=1)
    win = m.exact_window_for(arr, sp)
    twin = m.end_twins(arr, 1, angles)
    ts = []
    for rep in range(6):
        m.ctx.reset_best()
        m.ctx.set_option("near_window", win)
        m.ctx.match(arr, sp, sync=True)
        m.ctx.set_option("near_window", 0.0)
        t0 = time.perf_counter()
        st = m.ctx.settle_exact(twin, m.EXACT_MAX_F64)
        ts.append(time.perf_counter() - t0)
    m.ctx.profile(1)
    m.ctx.reset_best()
    m.ctx.set_option("near_window", win)
    m.ctx.match(arr, sp, sync=True)
    m.ctx.set_option("near_window", 0.0)
    st = m.ctx.settle_exact(twin, m.EXACT_MAX_F64)
    prof = m.ctx.profile_get()
    m.ctx.profile(0)
    print("scale %5.1f  bbox %s  settle %.3f ms (min of 6; device bracket %.3f ms)  %s" % (scale, bbox, 1e3 * min(ts), prof["k_settle"][1], st), flush=True)
