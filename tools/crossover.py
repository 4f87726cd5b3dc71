#!/usr/bin/env python3
"""Real-space vs FFT crossover: time both paths per template size (taps)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = synthetic.synthetic_scarp(n)
m = sl.Matcher(g)
if len(sys.argv) > 2:                                   # engine option "variant" (16: the 256 x 16 patch of the real-space kernel throughout)
    m.ctx.set_option("variant", float(sys.argv[2]))
angs = _plan.angle_grid()[[30, 90, 150]]
print("DEM %dx%d; per row: scale, age, taps(n), direct ms/template, fft ms/template, plan" % (n, n))
SCALES = tuple(int(x) for x in os.environ.get("CROSSOVER_SCALES", "5,10,20,50,100").split(","))
for scale in SCALES:
    for age in (1.0, 10.0, 100.0, 1000.0):
        row = []
        nn_last = 0
        dev_ms = float("nan")
        for meth in ("direct", "fft"):
            arr, bbox, area = m.describe(sl.Scarp, scale, np.array([age]), angs)
            if meth == "direct" and nn_last > 40000:         # (the largest supports take seconds in real space)
                row.append(float("nan")); continue
            m.search(sl.Scarp, scale, [age], angs, method=meth)      # warm
            t0 = time.perf_counter()
            for _ in range(3):
                m.search(sl.Scarp, scale, [age], angs, method=meth)
            row.append((time.perf_counter() - t0) / 3 / len(angs) * 1e3)
            if meth == "direct":                                      # the kernel alone (HIP events around every launch)
                m.ctx.profile(1)
                m.search(sl.Scarp, scale, [age], angs, method=meth)
                kn, kms = m.ctx.profile_get()["k_direct"]
                m.ctx.profile(0)
                dev_ms = kms / len(angs)
        nn, _ = m.ctx.template_sums(1)
        # FP32 rate of the real-space path on its FMA count: 2 FMA = 4 flop per tap and output cell
        tf = 4.0 * nn[0] * n * n / (row[0] * 1e-3) / 1e12 if row[0] == row[0] else float("nan")
        tfd = 4.0 * nn[0] * n * n / (dev_ms * 1e-3) / 1e12 if row[0] == row[0] else float("nan")
        print("scale %4d age %7.1f taps %7d  direct %8.3f ms (%5.1f TFLOP/s, %4.1f %% of 157; kernel alone %7.3f ms = %4.1f %%)  fft %8.3f  %s"
              % (scale, age, nn[0], row[0], tf, 100 * tf / 157.3, dev_ms, 100 * tfd / 157.3, row[1], m.plan))
