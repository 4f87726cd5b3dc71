#!/usr/bin/env python3
"""BASELINE configs C1 / C5 under engine option sets: step time and per-kernel device time (profiling brackets
on every launch).   python tools/small_lab.py C5 "dbg=0" "dbg=1" ...   (dbg: an SC_ABLATE build, SCARPLET_HIP_LIB)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan
cfg, sets = sys.argv[1], sys.argv[2:]
f = np.load(os.path.join(ROOT, "tests", "golden", "dem_carrizo.npz" if cfg == "C1" else "dem_grandcanyon.npz"))
g = sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"]))
m = sl.Matcher(g)
if cfg == "C1":
    lim = 17 * np.pi / 180
    jobs = [(sl.Scarp, 100.0, [10.0], _plan.angle_grid(-lim, lim))]
else:
    jobs = [(sl.Channel, s, [0.1], _plan.angle_grid()) for s in (5., 10., 20., 40., 80.)]
work = []
for cls, sc, par, ang in jobs:
    arr, bbox, area = m.describe(cls, sc, np.asarray(par), ang)
    plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(par))
    work.append((arr, sp))
for spec in sets:
    for kv in spec.split(","):
        try:
            m.ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
        except Exception as e:
            print("(option %s ignored: %s)" % (kv, e))
    for _ in range(3):
        for arr, sp in work:
            m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)
    m.ctx.profile(1)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        for arr, sp in work:
            m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)
    dt = (time.perf_counter() - t0) / reps
    prof = m.ctx.profile_get(); m.ctx.profile(0)
    print("%-10s step %.3f ms | %s" % (spec, 1e3 * dt, "  ".join("%s %.3f ms (%.1f us x %d)" % (k.replace("k_", ""), ms / reps, 1e3 * ms / max(n, 1), n // reps) for k, (n, ms) in prof.items() if n)), flush=True)
