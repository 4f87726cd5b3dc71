#!/usr/bin/env python3
"""Random searches against the ORACLE (float64 restatement of the reference): DEM size and parity, cell size, sign of dy,
template class, scale, parameters, orientations - through the FFT path, the real-space path, `method="auto"` and
`exact=True`.  Per case and path: cells outside the parity tolerance (oracle.PARITY with the path's tie window), cells
whose (age, angle) is not the oracle's own argmax, the measured amp / SNR error.  (The formal versions of these checks
are the `-m gpu` tests; this is the wide net.)
usage: python tools/fuzz_oracle.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scarplet_oracle as orc          # noqa: E402
import scarplet_amd as sl              # noqa: E402
from scarplet_amd import WindowedTemplate as WT   # noqa: E402

KIND = {WT.Scarp: orc.SCARP, WT.Ricker: orc.RICKER, WT.Channel: orc.RICKER,
        WT.RightFacingUpperBreakScarp: orc.RIGHT_UPPER, WT.LeftFacingUpperBreakScarp: orc.LEFT_UPPER}
CLASSES = [WT.Scarp, WT.Scarp, WT.Channel, WT.Ricker, WT.LeftFacingUpperBreakScarp, WT.RightFacingUpperBreakScarp]

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
ONLY = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]        # these cases only, with the cells that are off
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
tot = {}
t_start = time.time()
for case in range(n_cases):
    ny, nx = (int(v) for v in rng.integers(48, 420, size=2))
    if case % 7 == 0:
        ny, nx = int(2 ** rng.integers(6, 9)), int(2 ** rng.integers(6, 9))       # periodic power-of-two DEMs: no halo
    cls = CLASSES[int(rng.integers(0, len(CLASSES)))]
    de = float(rng.choice([1.0, 1.0, 2.0, 0.5]))
    scale = float(rng.uniform(4, min(ny, nx) / 4.5)) * de
    if cls in (WT.Channel, WT.Ricker):
        params = list(np.round(rng.uniform(0.05, 0.4, size=int(rng.integers(1, 4))) / de, 4))
    else:
        params = list(np.round(10 ** rng.uniform(0, 2.6, size=int(rng.integers(1, 7))) * de * de, 3))
    angles = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, size=int(rng.integers(1, 8))))
    if case % 5 == 0:
        angles = np.array([-np.pi / 2, 0.0, np.pi / 2])          # windows with a hole at xr = 0, the +-pi/2 twins
    z = (np.cumsum(np.cumsum(rng.standard_normal((ny, nx)), 0), 1) * 0.01 + rng.standard_normal((ny, nx)) * 0.05)
    if case % 4 == 1:
        z = np.round(z * 20.0).astype(np.int16)                  # integer DEMs (the Grand Canyon sample is int16)
    dy = -de if case % 3 == 0 else de
    if ONLY and case not in ONLY:
        continue
    g = sl.DEMGrid.from_array(z.astype(np.float32), de, dy)
    zz = g._griddata
    kind = KIND[cls]
    a_st, s_st = orc.snr_stack(zz, de, dy, kind, scale, params, angles)
    T = len(params) * len(angles)
    ages = np.repeat(np.asarray(params, float), len(angles))
    angs = np.tile(np.asarray(angles, float), len(params))
    A, S = a_st.reshape(T, ny, nx), s_st.reshape(T, ny, nx)
    tol = dict(amp_tol=(orc.PARITY["amp"][0], orc.PARITY["amp"][1] * float(np.max(np.abs(A)))),
               snr_tol=(orc.snr_tolerance(kind)[0], orc.snr_tolerance(kind)[1] * float(np.max(S))))
    line = "case %3d %4dx%-4d %-26s de %.1f dy %+.1f scale %6.1f %d x %d" % (case, ny, nx, cls.__name__, de, dy, scale,
                                                                           len(params), len(angles))
    for name, kw in (("fft", dict(method="fft")), ("direct", dict(method="direct")), ("auto", dict(method="auto")),
                     ("exact", dict(method="fft", exact=True)), ("exact-direct", dict(method="direct", exact=True)),
                     ("exact-auto", dict(method="auto", exact=True))):
        try:
            m = sl.Matcher(g)
            res = m.search(cls, scale, params, angles, **kw).result_array()
        except Exception as e:                                    # (a window the real-space slab does not hold, ...)
            line += "  | %s: %s" % (name, str(e)[:60])
            continue
        window = orc.tie_window("direct" if name in ("direct", "exact", "exact-direct", "exact-auto") else "fft", kind)
        chk = orc.check_fold(res, A, S, ages, angs, tie_rtol=window, **tol)
        t = tot.setdefault(name, dict(cases=0, cells=0, bad=0, inexact=0, snr_err=0.0, amp_err=0.0, worst=None))
        t["cases"] += 1
        t["cells"] += chk["n"]
        t["bad"] += chk["n_bad"]
        t["inexact"] += chk["n_inexact"]
        if chk["snr_err"] > t["snr_err"]:
            t["snr_err"], t["worst"] = chk["snr_err"], case
        t["amp_err"] = max(t["amp_err"], chk["amp_err"])
        line += "  | %s bad %d off-argmax %d err %.1e" % (name, chk["n_bad"], chk["n_inexact"], chk["snr_err"])
        if name.startswith("exact"):
            line += " (%s)" % getattr(m, "exact_stats", {}).get("route", "host")
        if ONLY and chk["n_bad"]:
            for (i, j) in np.argwhere(~chk["ok"])[:6]:
                hit = np.nonzero((ages == res[1][i, j]) & (angs == res[2][i, j]))[0]
                srt = np.argsort(S[:, i, j])[::-1][:3]
                line += "\n      %s BAD cell (%d,%d): carries amp %.9g age %g ang %.4f snr %.9g | oracle at the carried one %s | oracle top3 %s | max|A| %.4g max S %.4g" % (
                    name, i, j, res[0][i, j], res[1][i, j], res[2][i, j], res[3][i, j],
                    [(int(t_), "amp %.9g snr %.9g" % (A[t_, i, j], S[t_, i, j])) for t_ in hit],
                    [(int(t_), "%.9g" % S[t_, i, j]) for t_ in srt], float(np.max(np.abs(A))), float(np.max(S)))
        if ONLY and chk["n_inexact"]:
            smax = S.max(axis=0)
            carried = np.zeros((ny, nx), bool)
            for t_ in range(T):
                carried |= (res[1] == ages[t_]) & (res[2] == angs[t_]) & (S[t_] >= smax * (1 - 1e-9))
            below = (smax <= tol["snr_tol"][1]) & (np.abs(res[3]) <= tol["snr_tol"][1])
            for (i, j) in np.argwhere(~carried & ~below & (smax > 0))[:6]:
                srt = np.argsort(S[:, i, j])[::-1][:3]
                hit = np.nonzero((ages == res[1][i, j]) & (angs == res[2][i, j]))[0]
                line += "\n      %s cell (%d,%d): carries age %g ang %.4f snr %.9g | oracle top3 %s | oracle at the carried one %s | %s" % (
                    name, i, j, res[1][i, j], res[2][i, j], res[3][i, j],
                    [(int(t_), "%.9g" % S[t_, i, j]) for t_ in srt], [(int(t_), "%.9g" % S[t_, i, j]) for t_ in hit],
                    getattr(m, "exact_stats", None))
    print(line, flush=True)
print("\n%d cases in %.0f s" % (n_cases, time.time() - t_start))
for name, t in tot.items():
    print("%-7s %3d cases %9d cells: outside the tolerance %d, off the oracle's argmax %d, largest SNR error %.2e (case %s), amp %.2e"
          % (name, t["cases"], t["cells"], t["bad"], t["inexact"], t["snr_err"], t["worst"], t["amp_err"]))
sys.exit(1 if any(t["bad"] for t in tot.values()) else 0)
