#!/usr/bin/env python3
"""Random searches (DEM size and parity, cell size, scale, ages, orientations, template class) through the
real-space path (k_direct2, default form and variant 11) and through the FFT path: the share of cells
with the same (age, angle), the largest SNR difference relative to the map's maximum.
usage: python tools/fuzz_direct.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic, WindowedTemplate as WT

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
ctx = sl._lib.Context(0)
for case in range(n_cases):
    ny, nx = (int(v) for v in rng.integers(90, 1500, size=2))
    cls = [sl.Scarp, sl.Scarp, WT.Channel, WT.LeftFacingUpperBreakScarp, WT.RightFacingUpperBreakScarp][int(rng.integers(0, 5))]
    de = float(rng.choice([1.0, 1.0, 2.0, 0.5]))
    scale = float(rng.uniform(4, 40)) * de
    if cls is WT.Channel:
        params = list(rng.uniform(0.05, 0.4, size=int(rng.integers(1, 4))) / de)
    else:
        params = list(10 ** rng.uniform(0, 2.6, size=int(rng.integers(1, 8))) * de * de)
    angles = np.sort(rng.uniform(-np.pi / 2, np.pi / 2, size=int(rng.integers(1, 7))))
    if case % 5 == 0:
        angles = np.array([-np.pi / 2, 0.0, np.pi / 2])          # the windows with a hole at xr = 0
    g0 = synthetic.synthetic_scarp(nx, seed=100 + case, ny=ny)
    g = sl.DEMGrid.from_array(g0._griddata, de, -de if case % 3 == 0 else de)
    out = {}
    try:
        for name, method, variant in (("fft", "fft", 0), ("direct", "direct", 0), ("direct11", "direct", 11)):
            ctx.set_option("variant", variant)
            m = sl.Matcher(g, ctx=ctx)
            out[name] = m.search(cls, scale, params, angles, method=method).result_array()
    except Exception as e:
        print("case %d %dx%d %s scale %.1f: %s" % (case, ny, nx, cls.__name__, scale, e))
        bad += 1
        continue
    finally:
        ctx.set_option("variant", 0)
    f, d, d11 = out["fft"], out["direct"], out["direct11"]
    # (Scarp at -pi/2 and +pi/2 is ONE template up to the sign of W, Ricker the same template: equal SNRs to
    #  the last bits, either may win - counted as the same orientation, as oracle.check_fold does)
    hp = np.pi / 2
    same = float(np.mean((f[1] == d[1]) & ((f[2] == d[2]) | ((np.abs(f[2]) == hp) & (np.abs(d[2]) == hp)))))
    rel = float(np.abs(f[3] - d[3]).max() / max(f[3].max(), 1e-30))
    rel11 = float(np.abs(d11[3] - d[3]).max() / max(d[3].max(), 1e-30))
    ok = same >= 0.995 and rel <= 2e-3 and rel11 <= 1e-4
    bad += not ok
    print("case %2d %4dx%-4d de %.1f %-28s scale %5.1f  %d params x %d angles  same (age, angle) as fft %.4f  max dSNR/max %.1e  "
          "tap-by-tap form vs shared %.1e  %s" % (case, ny, nx, de, cls.__name__, scale, len(params), len(angles), same, rel, rel11,
                                                   "ok" if ok else "CHECK"))
print("cases to check:", bad)
sys.exit(1 if bad else 0)
