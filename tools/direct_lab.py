#!/usr/bin/env python3
"""Real-space kernel lab: device time of k_direct2 (profiling brackets) under engine option sets, per support size.
   python tools/direct_lab.py "variant=0" "variant=11" [--n 4096]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic
ap = argparse.ArgumentParser()
ap.add_argument("sets", nargs="+")
ap.add_argument("--n", type=int, default=4096)
a = ap.parse_args()
g = synthetic.synthetic_scarp(a.n)
m = sl.Matcher(g)
angs = _plan.angle_grid()[[30, 90, 150]]
for scale, age in ((5, 1.0), (10, 100.0), (20, 1000.0), (50, 100.0), (100, 1.0), (100, 100.0), (100, 1000.0)):
    line = "scale %3d age %6.0f" % (scale, age)
    for spec in a.sets:
        for kv in spec.split(","):
            m.ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        m.search(sl.Scarp, scale, [age], angs, method="direct")
        m.ctx.profile(1)
        for _ in range(3):
            m.search(sl.Scarp, scale, [age], angs, method="direct")
        prof = m.ctx.profile_get()
        m.ctx.profile(0)
        nl, ms = prof["k_direct"]
        taps = m.ctx.template_sums(1)[0][0]
        tf = 4.0 * taps * a.n * a.n / (ms / nl * 1e-3) / 1e12
        line += " | %s: %8.3f ms  %5.1f TFLOP/s (%4.1f %%)" % (spec, ms / nl, tf, 100 * tf / 157.3)
    print(line + "  taps %d" % taps, flush=True)
