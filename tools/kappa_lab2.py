#!/usr/bin/env python3
"""For the noise-free surfaces: how far above the oracle's float32 floor are the cells the FFT path gets wrong?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scarplet_oracle as orc
import scarplet_amd as sl
from scarplet_amd import _plan
from scarplet_amd import WindowedTemplate as WT
from scipy.special import erf

y, x = np.mgrid[-100:100, -100:100].astype(float) * 2.0
z = (-erf((-x * np.sin(0.6) + y * np.cos(0.6)) / (2 * np.sqrt(25.0)))).astype(np.float32)
dx = dy = 2.0
kind, cls, scale, params, angles = orc.SCARP, WT.Scarp, 20, [5.0, 25.0, 100.0], _plan.angle_grid()[::15]
ny, nx = z.shape
m = sl.Matcher(sl.DEMGrid.from_array(z, dx, dy))
for age in params:
    for ang in angles[::3]:
        amp, snr = m.match_template(cls, scale, age, ang, method="fft")
        curv = orc.directional_curvature(z, dx, dy, ang)
        W, lim, err = orc.template_arrays(kind, scale, age, ang, nx, ny, dx)
        a, s, det = orc.match_arrays(curv, W, lim, err, details=True)
        ts, n = det["template_sum"], det["n"]
        e = 5.9604644775390625e-08
        rms2, rms4 = np.sqrt(np.mean(curv ** 2)), np.sqrt(np.mean(curv ** 4))
        floor1 = e * n * rms4 + 2 * np.abs(det["xcorr"]) * e * np.sum(np.abs(W)) * rms2 / ts + (e * np.sum(np.abs(W)) * rms2) ** 2 / ts
        resid = det["T3"] - det["xcorr"] ** 2 / ts
        ratio = resid / floor1
        rel = np.abs(snr - s) / np.maximum(s, 1e-3 * s.max())
        badc = (rel > 2e-3) & (s > 2e-6 * s.max()) & ~lim
        if badc.any():
            r = ratio[badc]
            print("age %6.1f ang %+.2f: cells off by > 2e-3: %5d; their residual / (eps32 floor, k=1): min %.3g median %.3g max %.3g;"
                  " worst rel err %.3g at ratio %.3g; device/oracle there %.3g" % (
                      age, ang, badc.sum(), r.min(), np.median(r), r.max(), rel[badc].max(), ratio[badc][np.argmax(rel[badc])],
                      (snr[badc] / s[badc])[np.argmax(rel[badc])]), flush=True)
            # error vs ratio bins
            for lo, hi in ((0, 4), (4, 16), (16, 64), (64, 256), (256, 1024), (1024, 1e9)):
                sel = (ratio >= lo) & (ratio < hi) & (s > 2e-6 * s.max()) & ~lim
                if sel.any():
                    print("      ratio [%g, %g): %6d cells, median rel err %.2e, max %.2e" % (lo, hi, sel.sum(), np.median(rel[sel]), rel[sel].max()))
