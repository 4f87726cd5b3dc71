#!/usr/bin/env python3
"""What exact=True costs: wall time of sl.match with and without it on the int16 Grand Canyon DEM (Channel, one scale,
181 orientations: the flags cover a quarter of the DEM, the mode answers with the whole-DEM real-space search and
float64 for the cells that leaves) and on a noisy synthetic scarp DEM (Scarp, 12 ages x 37 orientations: a handful
of patches).  usage: python tools/exact_cost.py"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic
warnings.simplefilter("ignore")
if os.environ.get("EXACT_WINDOW_DIRECT"):                      # lab: another window for the real-space path's own flags
    sl.Matcher.EXACT_WINDOW_DIRECT = float(os.environ["EXACT_WINDOW_DIRECT"])
if os.environ.get("EXACT_MAX_F64"):
    sl.Matcher.EXACT_MAX_F64 = float(os.environ["EXACT_MAX_F64"])
f = np.load(os.path.join(ROOT, "tests", "golden", "dem_grandcanyon.npz"))
cases = [("grandcanyon 512 x 512, Channel f=0.1 scale 10, 1 x 181", sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"])),
          sl.Channel, 10.0, [0.1], _plan.angle_grid()),
         ("synthetic scarp 1500 x 1400 (sigma 0.05), Scarp scale 30, 12 x 37", synthetic.synthetic_scarp(1400, ny=1500, seed=3),
          sl.Scarp, 30.0, list(_plan.age_grid()[::3]), _plan.angle_grid()[::5])]
for name, g, cls, scale, params, angles in cases:
    for exact in (False, True):
        m = sl.Matcher(g)
        m.search(cls, scale, params, angles, method="fft", exact=exact).result()          # warm
        t0 = time.perf_counter()
        m.search(cls, scale, params, angles, method="fft", exact=exact)
        r = m.result()
        dt = time.perf_counter() - t0
        print("%-70s exact=%-5s %8.1f ms  %s" % (name, exact, 1e3 * dt, getattr(m, "exact_stats", "") if exact else ""))
