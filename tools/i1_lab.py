#!/usr/bin/env python3
"""Kernel lab: the full C3 search (sustained: a warm-up search, then timed ones) under a list of
engine option sets, in ONE process (the DEM is synthesised once).  Prints the step time and the
per-kernel breakdown per option set, and checks every set's best record against the first set's
(bit for bit) unless an ablation option ("dbg") makes its results meaningless.

  python tools/i1_lab.py "variant=0" "variant=8"
  SCARPLET_HIP_LIB=scarplet_amd/libscarplet_hip_ablate.so python tools/i1_lab.py "dbg=0" "dbg=4" ...
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("sets", nargs="+", help="option sets: name=value[,name=value...]")
ap.add_argument("--n", type=int, default=10000)
ap.add_argument("--angles", type=int, default=181)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--reset", default="variant=0", help="options restored before every set")
ap.add_argument("--c2", action="store_true", help="BASELINE config C2 instead: 2048^2, 10 ages x 91 orientations")
a = ap.parse_args()
if a.c2:
    a.n = 2048

g = synthetic.synthetic_scarp(a.n)
m = sl.Matcher(g)
ages = _plan.age_grid()
angs = _plan.angle_grid()[np.round(np.linspace(0, 180, a.angles)).astype(int)]
if a.c2:
    ages = ages[np.round(np.linspace(0, 34, 10)).astype(int)]
    angs = _plan.angle_grid(-np.pi / 4, np.pi / 4)
arr, bbox, area = m.describe(sl.Scarp, 100, ages, angs)
plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
units = a.n * a.n * len(ages) * len(angs) / 1e6
base = None
print("plan", plan, flush=True)


def apply(spec):
    for kv in spec.split(","):
        if kv:
            k, v = kv.split("=")
            try:
                m.ctx.set_option(k, float(v))
            except Exception as e:          # "dbg" does not exist outside an SC_ABLATE build
                print("   (option %s ignored: %s)" % (kv, e))


for spec in a.sets:
    apply(a.reset)
    if "dbg" in a.reset or any("dbg" in s for s in a.sets):
        apply("dbg=0")
    apply(spec)
    for _ in range(a.warmup):
        m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)
    m.ctx.profile(8)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)
    dt = (time.perf_counter() - t0) / a.steps
    prof = m.ctx.profile_get()
    m.ctx.profile(0)
    line = "  ".join("%s %.0f ms (%.1f us x %d)" % (k.replace("k_", ""), ms / a.steps, 1e3 * ms / max(n, 1), n // a.steps)
                     for k, (n, ms) in prof.items() if n)
    check = ""
    if "dbg" not in spec or spec.strip() == "dbg=0":
        best = m.ctx.get_best()
        if base is None:
            base = best
            check = "reference record"
        else:
            same = [bool(np.array_equal(x, y)) for x, y in zip(best, base)]
            check = "record identical to the first set: amp %s snr %s id %s" % tuple(same)
            if not all(same):
                d = np.abs(best[1] - base[1])
                check += "  (max |d snr| %.3g, cells differing %d)" % (d.max(), int((best[2] != base[2]).sum()))
    print("%-28s step %.3f s  %.0f Mpx.t/s | %s | %s" % (spec, dt, units / dt, line, check), flush=True)
