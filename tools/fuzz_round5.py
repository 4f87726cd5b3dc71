#!/usr/bin/env python3
"""Random searches on small and mid-size DEMs (where the round-5 launch forms act: batches of up to 256 templates,
row-pass launches of up to 255 in shares, template-split column passes, the half-wave-per-column kernel at column
length 512, the curvature mixed in the forward row pass) against the same search with every one of those forms
switched off, and - where no orientations ride in pairs - against one orientation per launch sequence.  The best
records must be equal in every bit.  usage: python tools/fuzz_round5.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic, WindowedTemplate as WT

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
OFF = {"variant": 17, "split_i1": 0, "split_fill": 2048, "batch_templ": 64}
OFF18 = dict(OFF, variant=18)
OFF20 = dict(OFF, variant=20)
bad = 0
for case in range(n_cases):
    ny, nx = (int(v) for v in rng.integers(120, 2600, size=2))
    cls = [sl.Scarp, sl.Scarp, WT.Channel, WT.LeftFacingUpperBreakScarp][int(rng.integers(0, 4))]
    scale = float(rng.uniform(6, 60))
    if cls is WT.Channel:
        params = list(rng.uniform(0.03, 0.3, size=int(rng.integers(1, 12))))
    else:
        params = list(10 ** rng.uniform(0, 3.0, size=int(rng.integers(1, 41))))
    angles = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, size=int(rng.integers(1, 31))))
    g = synthetic.synthetic_scarp(nx, seed=100 + case, ny=ny)
    out, plan, names = [], None, []
    forms = [("default", {}), ("forms off", OFF), ("off + four-column 512", OFF18), ("off + 64 per row launch", OFF20)]
    if len(params) > 1:
        forms.append(("one orientation per sequence", {"batch": 0}))
    try:
        for name, opts in forms:
            ctx = sl._lib.Context(0)
            for k, v in opts.items():
                ctx.set_option(k, v)
            m = sl.Matcher(g, ctx=ctx)
            m.search(cls, scale, params, angles, method="fft")
            out.append(m.ctx.get_best())
            names.append(name)
            plan = m.plan
            del m
            ctx.close()
    except Exception as e:
        print("case %d %dx%d %s scale %.1f: %s" % (case, ny, nx, cls.__name__, scale, e))
        bad += 1
        continue
    diff = [names[k + 1] for k, o in enumerate(out[1:]) if not all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[0], o))]
    bad += bool(diff)
    print("case %2d %4dx%-4d %-26s scale %5.1f  %2d params x %2d angles  %s  %s"
          % (case, ny, nx, cls.__name__, scale, len(params), len(angles), plan, "identical" if not diff else "DIFFERENT: " + ", ".join(diff)))
print("cases with differences or errors:", bad)
sys.exit(1 if bad else 0)
