#!/usr/bin/env python3
"""Where does the float32 resolution floor of the FFT epilogue bind?  Noise-free surfaces,
FFT path at several kappa and the real-space path, against the oracle."""
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
z1 = (-erf((-x * np.sin(0.6) + y * np.cos(0.6)) / (2 * np.sqrt(25.0)))).astype(np.float32)
y, x = np.mgrid[-128:128, -128:128].astype(float)
d = -x * np.sin(-0.4) + y * np.cos(-0.4)
z2 = (-np.exp(-(d / 6.0) ** 2)).astype(np.float32)
y, x = np.mgrid[-300:300, -330:330].astype(float)
z3 = (-erf((-x * np.sin(1.1) + y * np.cos(1.1)) / (2 * np.sqrt(10.0))) + 0.01 * x).astype(np.float32)
cases = [("scarp de=2 s=20", z1, 2.0, 2.0, WT.Scarp, orc.SCARP, 20, [5.0, 25.0, 100.0], _plan.angle_grid()[::15]),
         ("channel", z2, 1.0, -1.0, WT.Channel, orc.RICKER, 10, [0.05, 0.1], _plan.angle_grid()[::12]),
         ("scarp+ramp 600x660", z3, 1.0, 1.0, WT.Scarp, orc.SCARP, 40, [3.0, 10.0, 30.0], _plan.angle_grid()[5::30])]
P = orc.PARITY
for (name, z, dx, dy, cls, kind, scale, params, angles) in cases:
    a_st, s_st = orc.snr_stack(z, dx, dy, kind, scale, params, angles, workers=8)
    T = len(params) * len(angles)
    ny, nx = z.shape
    A, S = a_st.reshape(T, ny, nx), s_st.reshape(T, ny, nx)
    smax = S.max(0)
    print("%s: oracle max SNR %.3g, median of per-cell max %.3g" % (name, smax.max(), np.median(smax)))
    for label, method, kappa in (("fft k=4", "fft", 4.0), ("fft k=1", "fft", 1.0), ("fft k=0", "fft", 0.0),
                                 ("direct", "direct", 4.0)):
        ctx = sl._lib.Context(0)
        ctx.set_option("kappa", kappa)
        m = sl.Matcher(sl.DEMGrid.from_array(z, dx, dy), ctx=ctx)
        res = m.search(cls, scale, params, angles, method=method).result()
        chk = orc.check_fold(res, A, S, np.repeat(params, len(angles)), np.tile(angles, len(params)),
                             tie_rtol=P["tie_rtol"], amp_tol=(P["amp"][0], P["amp"][1] * np.abs(A).max()),
                             snr_tol=(P["snr"][0], P["snr"][1] * S.max()))
        bad = ~chk["ok"]
        snr = np.asarray(res[3])
        hi = smax > 1e-3 * smax.max()
        print("   %-8s bad %6d of %d (exact %.4f); bad among cells with max SNR > 1e-3 of the map's: %d of %d; "
              "device/oracle SNR on bad cells: median %.3g, max %.3g" % (
                  label, bad.sum(), bad.size, chk["exact_frac"], (bad & hi).sum(), hi.sum(),
                  np.median(snr[bad] / smax[bad]) if bad.any() else 0, np.max(snr[bad] / smax[bad]) if bad.any() else 0),
              flush=True)
        del m; ctx.close()
