#!/usr/bin/env python3
"""Developer timing: one search on a synthetic DEM, per-kernel breakdown."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--ages", type=int, default=35)
ap.add_argument("--angles", type=int, default=5)
ap.add_argument("--method", default="fft")
ap.add_argument("--group", type=int, default=0)
ap.add_argument("--scale", type=float, default=100.0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--prof", type=int, default=8)
ap.add_argument("--variant", type=int, default=0, help="sc_set_option variant (include/scarplet_hip.h)")
ap.add_argument("--opt", default="", help="engine options name=value[,name=value...] (sc_set_option)")
a = ap.parse_args()
t0 = time.time()
g = synthetic.synthetic_scarp(a.n)
print("dem %dx%d built in %.1fs" % (a.n, a.n, time.time() - t0))
m = sl.Matcher(g)
m.ctx.set_option("variant", a.variant)
for kv in a.opt.split(","):
    if kv:
        m.ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
ages = _plan.age_grid()[np.round(np.linspace(0, 34, a.ages)).astype(int)]
angs = _plan.angle_grid()[np.round(np.linspace(0, 180, a.angles)).astype(int)]
for rep in range(a.reps):
    m.ctx.profile(a.prof)
    t0 = time.time()
    m.search(sl.Scarp, a.scale, ages, angs, method=a.method, group=a.group or None)
    dt = time.time() - t0
    work = a.n * a.n * len(ages) * len(angs) / 1e6
    print("rep %d: %.3fs  %.0f Mpx.tmpl/s  plan %s  mem %.2f GB" % (rep, dt, work / dt, m.plan, m.ctx.device_bytes() / 1e9))
    for k, (n, ms) in m.ctx.profile_get().items():
        if n:
            print("   %-12s launches %7d  total %9.1f ms  avg %8.1f us" % (k, n, ms, 1e3 * ms / n))
