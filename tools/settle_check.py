#!/usr/bin/env python3
"""sc_settle_exact (round 6) on an MI355X: the device settle against round 5's host routes on small searches (same (age,
orientation) everywhere), then its cost on the headline workload (C3) - exact=False / exact=True wall times of search +
result, and the settle alone.
usage: python tools/settle_check.py [c3]"""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic

warnings.simplefilter("ignore")


def run(g, cls, scale, params, angles, method, events):
    m = sl.Matcher(g)
    m.EXACT_USE_EVENTS = events
    t0 = time.perf_counter()
    m.search(cls, scale, params, angles, method=method, exact=True)
    r = np.stack(m.result())
    return r, time.perf_counter() - t0, dict(m.exact_stats), m.method_used


if len(sys.argv) < 2 or sys.argv[1] != "c3only":
    f = np.load(os.path.join(ROOT, "tests/golden/dem_grandcanyon.npz"))
    cases = [("grandcanyon Channel 1 x 181 fft", sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"])), sl.Channel, 10.0, [0.1],
              _plan.angle_grid(), "fft"),
             ("synthetic 1500 x 1400 Scarp 12 x 37 fft", synthetic.synthetic_scarp(1400, ny=1500, seed=3), sl.Scarp, 30.0, list(_plan.age_grid()[::3]),
              _plan.angle_grid()[::5], "fft"),
             ("synthetic 700 x 640 Scarp 12 x 181 fft (end twins)", synthetic.synthetic_scarp(640, ny=700, seed=5), sl.Scarp, 20.0, list(_plan.age_grid()[::3]),
              _plan.angle_grid(), "fft"),
             ("synthetic 300 x 280 Scarp 6 x 181 direct", synthetic.synthetic_scarp(280, ny=300, seed=7), sl.Scarp, 10.0, list(_plan.age_grid()[::6]),
              _plan.angle_grid(), "direct")]
    for name, g, cls, scale, params, angles, method in cases:
        r0 = np.stack(sl.Matcher(g).search(cls, scale, params, angles, method=method, exact=False).result())
        rd, td, sd, md = run(g, cls, scale, params, angles, method, True)
        rd, td, sd, md = run(g, cls, scale, params, angles, method, True)
        rh, th, sh, mh = run(g, cls, scale, params, angles, method, False)
        # the two end orientations are one template: compare the angle modulo that
        ang_d, ang_h = rd[2].copy(), rh[2].copy()
        for a in (ang_d, ang_h):
            a[np.abs(a - np.pi / 2) < 1e-12] = -np.pi / 2
        same = (rd[1] == rh[1]) & (ang_d == ang_h)
        a0 = r0[2].copy(); a0[np.abs(a0 - np.pi / 2) < 1e-12] = -np.pi / 2
        moved = ((rd[1] != r0[1]) | (ang_d != a0)).sum()
        print("%-52s device %.1f ms (%s) %s" % (name, 1e3 * td, md, sd))
        print("%-52s host   %.1f ms (%s) %s" % ("", 1e3 * th, mh, sh))
        print("%-52s cells where the two routes differ in (age, orientation): %d of %d; device route moved %d cells off the float32 answer; "
              "max |d snr| on equal cells %.2e" % ("", int((~same).sum()), same.size, int(moved),
                                                   float(np.max(np.abs(rd[3][same] - rh[3][same]) / np.maximum(rh[3][same], 1e-300)))), flush=True)

if len(sys.argv) > 1 and sys.argv[1] in ("c3", "c3only"):
    n = 10000
    g = synthetic.synthetic_scarp(n)
    ages, angles = _plan.age_grid(), _plan.angle_grid()
    m = sl.Matcher(g)
    for exact in (False, True, True, False, True):
        t0 = time.perf_counter()
        m.search(sl.Scarp, 100, ages, angles, method="fft", exact=exact)
        m.ctx.sync()
        t1 = time.perf_counter()
        res = m.result()
        t2 = time.perf_counter()
        print("C3 exact=%s: search %.3f s, result %.3f s  %s" % (exact, t1 - t0, t2 - t1, m.exact_stats if exact else ""), flush=True)
        del res
    # the settle alone: the search with the flags on, then the call
    arr, bbox, area = m.describe(sl.Scarp, 100, ages, angles)
    plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
    for rep in range(2):
        m.ctx.reset_best()
        m.ctx.set_option("near_window", m.EXACT_WINDOW[0])
        t0 = time.perf_counter()
        m.ctx.match(arr, sp, sync=True)
        t1 = time.perf_counter()
        m.ctx.set_option("near_window", 0.0)
        st = m.ctx.settle_exact(len(ages), 0.0)
        t2 = time.perf_counter()
        print("C3 near-tie search %.3f s, sc_settle_exact %.3f s  %s" % (t1 - t0, t2 - t1, st), flush=True)
