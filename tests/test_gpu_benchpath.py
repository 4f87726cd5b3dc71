"""The BENCHMARKED configuration against the oracle (BASELINE config C3).

bench.py runs the 10000 x 10000 synthetic DEM with all 35 ages per inverse
launch (group = 35), 2048 x 2048 tiles in pairs, several tile pairs per I2
launch.  This test runs exactly that plan (the planner's default for the
search, no overrides) on a spread of orientations and checks windows of the
result against oracle.snr_stack_window: inside a tile, across the corner
where four tiles (two tile pairs, two tile rows) meet, on the DEM's wrap
edges and in the partial last tiles.  The exact-argmax fraction and the
measured SNR error are printed and asserted (run with -s to see them)."""
import numpy as np
import pytest

import scarplet_oracle as orc
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic

pytestmark = pytest.mark.gpu
P = orc.PARITY


def check_window(res, z, kind, scale, ages, angles, win, margin, pool):
    a_st, s_st = orc.snr_stack_window(z, 1.0, 1.0, kind, scale, ages, angles, win, margin, pool=pool)
    T = len(ages) * len(angles)
    i0, i1, j0, j1 = win
    sub = tuple(np.asarray(r)[i0:i1, j0:j1] for r in res)
    return orc.check_fold(sub, a_st.reshape(T, i1 - i0, j1 - j0), s_st.reshape(T, i1 - i0, j1 - j0),
                          np.repeat(ages, len(angles)), np.tile(angles, len(ages)),
                          tie_rtol=orc.tie_window("fft", kind), amp_tol=(P["amp"][0], P["amp"][1] * np.max(np.abs(a_st))),
                          snr_tol=(P["snr"][0], P["snr"][1] * np.max(s_st)))


def test_bench_plan_windows_against_oracle(gpu_ctx, oracle_pool):
    n = 10000
    g = synthetic.synthetic_scarp(n)
    ages = _plan.age_grid()                                     # all 35: one inverse launch each
    angles = _plan.angle_grid()[[0, 23, 45, 90, 135, 157, 180]]  # -90, -67, -45, 0, 45, 67, 90 degrees
    m = sl.Matcher(g, ctx=gpu_ctx)
    m.search(sl.Scarp, 100, ages, angles, method="fft")
    p = m.plan
    # the plan bench.py runs with
    assert (p.Ty, p.Tx, p.nty, p.ntx, p.group) == (2048, 2048, 6, 6, 35), p
    res = m.result()
    V = p.Vy
    wins = {
        "tile interior": (800, 896, 900, 996),
        "corner of 4 tiles / 2 tile pairs": (V - 48, V + 48, V - 48, V + 48),
        "seam between tile rows 2|3, pair seam": (3 * V - 48, 3 * V + 48, 2 * V - 48, 2 * V + 48),
        "wrap corner": (0, 96, n - 96, n),
        "left wrap edge": (5000, 5096, 0, 96),
        "partial last tiles (bottom right)": (n - 400, n - 304, n - 420, n - 324),
    }
    worst = 0.0
    for name, win in wins.items():
        chk = check_window(res, g._griddata, orc.SCARP, 100, ages, angles, win, 160, oracle_pool)
        print("bench-plan window %-40s bad=%d inexact=%d below=%d exact=%.6f strict=%d tie=%d of %d  snr_err=%.2e amp_err=%.2e"
              % (name, chk["n_bad"], chk["n_inexact"], chk["n_below_only"], chk["exact_frac"], chk["n_strict"],
                 chk["n_tie"], chk["n"], chk["snr_err"], chk["amp_err"]))
        assert chk["n_bad"] == 0, (name, chk["n_bad"])
        assert chk["n_inexact"] == 0, (name, chk["n_inexact"])       # the benchmark DEM has a noise floor: exact, as an integer
        worst = max(worst, chk["snr_err"])
    # the tie window is meant to be twice the measured error
    assert worst <= 0.5 * orc.tie_window("fft", orc.SCARP), worst


def test_c3_full_grid_windows_against_oracle(gpu_ctx, oracle_pool):
    """BASELINE config C3 in full - the 10000 x 10000 DEM, all 35 ages x 181 orientations = 6335
    templates, the search bench.py times - against the oracle on windows that cover what can go
    wrong at full size: a tile interior, the corner where four tiles (two tile pairs) meet, a
    seam between tile rows, the wrap corner of the periodic DEM, the partial last tiles, the
    window-limit border.  Every window meets ALL 6335 templates (oracle.snr_stack_window).  The
    reference pins the full grid with synthetic_match1.npy (tests/test_core.py:28-45); this is
    the same pin at the benchmark's size.  Then the search is repeated: the record must be the
    same in every bit."""
    n = 10000
    g = synthetic.synthetic_scarp(n)
    ages, angles = _plan.age_grid(), _plan.angle_grid()
    m = sl.Matcher(g, ctx=gpu_ctx)
    m.search(sl.Scarp, 100, ages, angles, method="fft")
    p = m.plan
    assert (p.nty, p.ntx, p.group) == (6, 6, 35) and p.Ty == 2048, p
    res = m.result()
    best0 = m.ctx.get_best()
    V, w = p.Vy, 48
    wins = {
        "tile interior": (800, 900),
        "corner of 4 tiles / 2 tile pairs": (V - 24, V - 24),
        "seam between tile rows 2|3": (3 * V - 24, 4000),
        "wrap corner": (0, n - w),
        "partial last tiles": (n - 400, n - 420),
        "window-limit border (top)": (150, 6000),
    }
    worst, cells, ties = 0.0, 0, 0
    for name, (i0, j0) in wins.items():
        chk = check_window(res, g._griddata, orc.SCARP, 100, ages, angles, (i0, i0 + w, j0, j0 + w), 160, oracle_pool)
        print("C3 full grid, window %-34s bad=%d inexact=%d below=%d exact=%.6f strict=%d tie=%d of %d  snr_err=%.2e amp_err=%.2e"
              % (name, chk["n_bad"], chk["n_inexact"], chk["n_below_only"], chk["exact_frac"], chk["n_strict"],
                 chk["n_tie"], chk["n"], chk["snr_err"], chk["amp_err"]))
        assert chk["n_bad"] == 0, (name, chk["n_bad"])
        assert chk["n_inexact"] == 0, (name, chk["n_inexact"])
        worst, cells, ties = max(worst, chk["snr_err"]), cells + chk["n"], ties + chk["n_tie"]
    print("C3 full grid: %d cells x %d templates, 0 cells off the oracle's argmax (%d with a second candidate inside "
          "the tie window), largest SNR error %.2e" % (cells, len(ages) * len(angles), ties, worst))
    assert worst <= 0.5 * orc.tie_window("fft", orc.SCARP), worst
    m.search(sl.Scarp, 100, ages, angles, method="fft")
    b = m.ctx.get_best()
    same = [bool(np.array_equal(x.view(np.uint32), y.view(np.uint32))) for x, y in zip(b, best0)]
    print("C3 full grid repeated: record identical in every bit: amp %s snr %s id %s" % tuple(same))
    assert all(same)
