#!/usr/bin/env python3
"""Random searches (DEM size, scale, ages, orientations, template class) run through the default
kernels and through the cross-check variants of the inverse column pass; the best records must be
equal in every bit.  usage: python tools/fuzz_variants.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic, WindowedTemplate as WT

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    ny, nx = (int(v) for v in rng.integers(300, 4200, size=2))
    cls = [sl.Scarp, sl.Scarp, WT.Channel, WT.LeftFacingUpperBreakScarp][int(rng.integers(0, 4))]
    scale = float(rng.uniform(8, 110))
    if cls is WT.Channel:
        params = list(rng.uniform(0.02, 0.3, size=int(rng.integers(1, 9))))
    else:
        params = list(10 ** rng.uniform(0, 3.4, size=int(rng.integers(1, 20))))
    angles = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, size=int(rng.integers(1, 6))))
    g = synthetic.synthetic_scarp(nx, seed=case, ny=ny)
    out, plan = [], None
    try:
        for variant in (0, 2, 6):
            ctx = sl._lib.Context(0)
            ctx.set_option("variant", variant)
            m = sl.Matcher(g, ctx=ctx)
            m.search(cls, scale, params, angles, method="fft")
            out.append(m.ctx.get_best())
            plan = m.plan
            del m
            ctx.close()
    except Exception as e:
        print("case %d %dx%d %s scale %.1f: %s" % (case, ny, nx, cls.__name__, scale, e))
        continue
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for o in out[1:] for a, b in zip(out[0], o))
    bad += not same
    print("case %2d %4dx%-4d %-26s scale %5.1f  %2d params x %d angles  %s  %s"
          % (case, ny, nx, cls.__name__, scale, len(params), len(angles), plan, "identical" if same else "DIFFERENT"))
print("cases with differences:", bad)
sys.exit(1 if bad else 0)
