#!/usr/bin/env python3
"""Developer probe: run the HIP paths against the oracle and print errors.
(The formal versions of these checks are the `-m gpu` tests.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scarplet_oracle as orc          # noqa: E402
import scarplet_amd as sl              # noqa: E402
from scarplet_amd import _plan         # noqa: E402
from scarplet_amd import WindowedTemplate as WT   # noqa: E402

rng = np.random.default_rng(11)
KIND = {WT.Scarp: orc.SCARP, WT.Ricker: orc.RICKER, WT.Channel: orc.RICKER,
        WT.RightFacingUpperBreakScarp: orc.RIGHT_UPPER,
        WT.LeftFacingUpperBreakScarp: orc.LEFT_UPPER}


def dem(ny, nx, de=1.0, dy=None):
    z = (np.cumsum(np.cumsum(rng.standard_normal((ny, nx)), 0), 1) * 0.01
         + rng.standard_normal((ny, nx)) * 0.05).astype(np.float32)
    return sl.DEMGrid.from_array(z, de, dy)


def relerr(a, b):
    return float(np.max(np.abs(a - b) / (np.abs(b) + 1e-3 * np.max(np.abs(b)) + 1e-30)))


def check_curvature():
    g = dem(70, 90, 2.0)
    m = sl.Matcher(g)
    for ang in (0.0, 0.4, -np.pi / 2):
        cc, sc2, ss = _plan.curvature_coefficients(ang)
        c = m.ctx.curvature(cc, sc2, ss, (70, 90))
        o = orc.directional_curvature(g._griddata, 2.0, 2.0, ang)
        print("curvature ang=%.2f  max|d|=%.3g (scale %.3g)" % (ang, np.max(np.abs(c - o)), np.max(np.abs(o))))


def check_single(cls, ny, nx, de, scale, par, ang, methods=("direct", "fft"), dy=None):
    g = dem(ny, nx, de, dy)
    z = g._griddata
    o_amp, _, _, o_snr, det = orc.match_template(z, de, g._georef_info.dy, KIND[cls], scale, par, ang, details=True)
    m = sl.Matcher(g)
    for meth in methods:
        t0 = time.time()
        amp, snr = m.match_template(cls, scale, par, ang, method=meth)
        n, ts = m.ctx.template_sums(1)
        print("%-8s %-6s %dx%d de=%g s=%g p=%g a=%.2f  amp %.2e snr %.2e  n %g/%g ts %.3e  plan %s  %.2fs" % (
            cls.__name__[:8], meth, ny, nx, de, scale, par, ang, relerr(amp, o_amp), relerr(snr, o_snr),
            n[0], det["n"], abs(ts[0] - det["template_sum"]) / det["template_sum"],
            m.plan if meth == "fft" else "-", time.time() - t0))


def check_fold(cls, ny, nx, de, scale, params, angs, method):
    g = dem(ny, nx, de)
    z = g._griddata
    m = sl.Matcher(g)
    t0 = time.time()
    m.search(cls, scale, params, angs, method=method)
    res = m.result()
    dt = time.time() - t0
    a_st, s_st = orc.snr_stack(z, de, de, KIND[cls], scale, params, angs)
    T = len(params) * len(angs)
    ages = np.repeat(np.asarray(params, float), len(angs))
    angles = np.tile(np.asarray(angs, float), len(params))
    chk = orc.check_fold(res, a_st.reshape(T, ny, nx), s_st.reshape(T, ny, nx), ages, angles,
                         tie_rtol=1e-4, amp_tol=(2e-4, 2e-6 * float(np.max(np.abs(a_st)))),
                         snr_tol=(2e-3, 2e-6 * float(np.max(s_st))))
    print("fold %-8s %-6s %dx%d %d templates: bad=%d strict=%d tie=%d of %d  (%.2fs)" % (
        cls.__name__[:8], method, ny, nx, T, chk["n_bad"], chk["n_strict"], chk["n_tie"], chk["n"], dt))
    bad = np.argwhere(~chk["ok"])[:6]
    S = s_st.reshape(T, ny, nx)
    A = a_st.reshape(T, ny, nx)
    for (i, j) in bad:
        srt = np.argsort(S[:, i, j])[::-1][:3]
        hit = np.nonzero((ages == res[1][i, j]) & (angles == res[2][i, j]))[0]
        print("   bad (%d,%d): got amp=%.6g age=%g ang=%.4f snr=%.6g | oracle top3 %s | oracle at chosen: %s" % (
            i, j, res[0][i, j], res[1][i, j], res[2][i, j], res[3][i, j],
            [(int(t), float("%.6g" % S[t, i, j]), float("%.4g" % A[t, i, j])) for t in srt],
            [(int(t), float("%.6g" % S[t, i, j]), float("%.4g" % A[t, i, j])) for t in hit]))
    return chk


if __name__ == "__main__":
    check_curvature()
    check_single(WT.Scarp, 64, 64, 1.0, 10, 10.0, 0.3)
    check_single(WT.Scarp, 61, 75, 1.0, 8, 3.2, -1.2)
    check_single(WT.Scarp, 80, 64, 2.0, 20, 31.6, np.pi / 2)
    check_single(WT.Scarp, 65, 65, 1.0, 10, 1.0, 0.0)
    check_single(WT.Scarp, 200, 180, 1.0, 12, 5.0, 0.5)
    check_single(WT.Scarp, 300, 300, 1.0, 100, 100.0, 0.7)
    check_single(WT.Scarp, 512, 512, 1.0, 50, 31.6, -0.4)
    check_single(WT.Channel, 64, 72, 1.0, 5, 0.1, 0.8, dy=-1.0)
    check_single(WT.Ricker, 63, 64, 1.0, 8, 0.2, -0.3)
    check_single(WT.Ricker, 256, 256, 1.0, 10, 0.1, 0.3)
    check_single(WT.RightFacingUpperBreakScarp, 64, 66, 1.0, 10, 10.0, 0.2)
    check_single(WT.LeftFacingUpperBreakScarp, 61, 64, 1.0, 10, 5.0, -0.6)
    check_single(WT.Scarp, 1100, 1000, 1.0, 100, 1000.0, 0.6, methods=("fft",))
    check_single(WT.Scarp, 1000, 1500, 1.0, 100, 300.0, -0.8, methods=("fft",))
    check_single(WT.Scarp, 2500, 2300, 1.0, 100, 100.0, 0.4, methods=("fft",))
    check_single(WT.Scarp, 2200, 2100, 1.0, 100, 30.0, -0.2, methods=("fft",))      # 3x3 tiles: odd count
    for meth in ("direct", "fft"):
        check_fold(WT.Scarp, 96, 90, 1.0, 10, [1.0, 3.16, 10.0, 31.6], _plan.angle_grid(-0.5, 0.5), meth)
        check_fold(WT.Channel, 80, 96, 1.0, 6, [0.1, 0.2], _plan.angle_grid(-np.pi / 2, np.pi / 2)[::6], meth)
    # odd number of templates per orientation (paired-template mode leaves one single)
    check_fold(WT.Scarp, 96, 90, 1.0, 10, [1.0, 3.16, 10.0, 31.6, 50.0], _plan.angle_grid(-0.5, 0.5)[::4], "fft")
    check_fold(WT.Scarp, 130, 140, 1.0, 12, [2.0], _plan.angle_grid(-0.5, 0.5)[::4], "fft")
    # odd tile count (the last tile alone in its pair) and several templates
    check_fold(WT.Scarp, 2200, 2100, 1.0, 100, [3.0, 30.0, 300.0], np.array([-0.7, 1.2]), "fft")
