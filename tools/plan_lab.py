#!/usr/bin/env python3
"""BASELINE config C2 (2048^2, 10 ages x 91 orientations) under alternative tile plans: the planner's
(one circular 2048 x 2048 tile: templates ride in pairs through the four-wave column pass) against
column length 1024 (three tiles along y, circular along x: a tile pair on the eight-wave column pass
plus one paired-template tile at 1024).   python tools/plan_lab.py [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, _lib, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
g = synthetic.synthetic_scarp(n)
m = sl.Matcher(g)
if n == 2048:
    ages = _plan.age_grid()[np.round(np.linspace(0, 34, 10)).astype(int)]
    angs = _plan.angle_grid(-np.pi / 4, np.pi / 4)
else:                                   # the C3 grid on a few orientations
    ages = _plan.age_grid()
    angs = _plan.angle_grid()[::20]
arr, bbox, area = m.describe(sl.Scarp, 100, ages, angs)
p0, sp0 = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
pmin, pmax, qmin, qmax = bbox


def plan(Ty, circ_y, Tx, circ_x):
    Vy = Ty if circ_y else Ty - (pmax - pmin)
    Vx = Tx if circ_x else Tx - (qmax - qmin)
    nty = 1 if circ_y else -(-n // Vy)
    ntx = 1 if circ_x else -(-n // Vx)
    circ_y, circ_x = circ_y and n == Ty, circ_x and n == Tx
    Vy = Ty if circ_y else Ty - (pmax - pmin)
    Vx = Tx if circ_x else Tx - (qmax - qmin)
    nty = 1 if circ_y else -(-n // Vy)
    ntx = 1 if circ_x else -(-n // Vx)
    return _lib.sc_plan(method=1, Ty=Ty, Tx=Tx, Vy=Vy, Vx=Vx, nty=nty, ntx=ntx, circ_y=int(circ_y), circ_x=int(circ_x),
                        Py=Ty // 2 if circ_y else pmax, Qx=Tx // 2 if circ_x else qmax, group=len(ages))


base = None
for name, sp in (("planner", sp0), ("2048 x 2048", plan(2048, True, 2048, True)), ("1024 x 2048", plan(1024, False, 2048, True)),
                 ("2048 x 1024", plan(2048, True, 1024, False)), ("1024 x 1024", plan(1024, False, 1024, False)),
                 ("planner again", sp0)):
    for _ in range(2):
        m.ctx.forget_spectra(); m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)
    m.ctx.profile(4)
    t0 = time.perf_counter()
    for _ in range(steps):
        m.ctx.forget_spectra(); m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)
    dt = (time.perf_counter() - t0) / steps
    prof = m.ctx.profile_get(); m.ctx.profile(0)
    best = m.ctx.get_best()
    if base is None:
        base = best
    same_id = float((best[2] == base[2]).mean())
    dsnr = float(np.max(np.abs(best[1] - base[1]) / np.maximum(base[1], 1e-3 * base[1].max())))
    print("%-24s tiles %dx%d of %dx%d  %.2f ms  | %s | same id %.6f, max rel d snr %.1e"
          % (name, sp.nty, sp.ntx, sp.Ty, sp.Tx, 1e3 * dt,
             "  ".join("%s %.2f" % (k.replace("k_", ""), ms / steps) for k, (c, ms) in prof.items() if c), same_id, dsnr), flush=True)
