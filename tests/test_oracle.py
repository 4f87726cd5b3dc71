"""The oracle against the reference's golden vectors (tests/golden, produced
by oracle/gen_golden.py from the unmodified reference)."""
import numpy as np
import pytest

import scarplet_oracle as orc
from conftest import golden, load_cases


def test_templates_reference_goldens():
    # scarplet/tests/test_WindowedTemplate.py:26-32, 51-57
    assert np.allclose(orc.scarp_template(100, 10, 0, 100, 100, 1),
                       np.load(golden("ref_scarp_template.npy")))
    assert np.allclose(orc.ricker_template(100, 0.1, 0, 100, 100, 1),
                       np.load(golden("ref_channel_template.npy")))


def test_templates_reference_classes():
    for c in load_cases("ref_templates.npz"):
        W, lim, err = orc.template_arrays(str(c["kind"]), float(c["d"]), float(c["p"]),
                                          float(c["ang"]), int(c["nx"]), int(c["ny"]),
                                          float(c["de"]))
        assert np.allclose(W, c["W"], rtol=1e-13, atol=0)
        assert np.array_equal(W != 0, c["W"] != 0)
        assert np.array_equal(lim, c["lim"])
        if c["err"].size:
            assert np.array_equal(err, c["err"])


def test_curvature_faultzone_goldens():
    # scarplet/tests/test_dem.py:32-45, on corner crops of faultzone.tif
    f = np.load(golden("ref_faultzone_curvature.npz"))
    for deg in f["angles_deg"]:
        a = np.radians(float(deg))
        if deg == -90:
            a = -np.pi / 2
        elif deg == 90:
            a = np.pi / 2
        elif deg == -45:
            a = -np.pi / 4
        elif deg == 45:
            a = np.pi / 4
        tl = orc.directional_curvature(f["z_tl"], 2.0, 2.0, a)
        br = orc.directional_curvature(f["z_br"], 2.0, 2.0, a)
        # the crop's interior edges see a different stencil than the full image
        assert np.allclose(tl[:-1, :-1], f["gold_tl_%d" % deg][:-1, :-1])
        assert np.allclose(br[1:, 1:], f["gold_br_%d" % deg][1:, 1:])


def test_match_template_reference():
    for c in load_cases("ref_match_template.npz"):
        amp, _, _, snr, det = orc.match_template(
            c["z"], float(c["dx"]), float(c["dy"]), str(c["kind"]), float(c["scale"]),
            float(c["age"]), float(c["ang"]), details=True)
        assert np.allclose(amp, c["amp"], rtol=1e-9, atol=1e-13)
        assert np.allclose(snr, c["snr"], rtol=1e-7, atol=1e-10)
        assert abs(det["n"] - float(c["n"])) < 1e-9
        assert np.isclose(det["template_sum"], float(c["ts"]), rtol=1e-13)


def test_fold_semantics_reference():
    f = np.load(golden("ref_fold.npz"))
    res = orc.compare(((f["amps"][i], f["ages"][i], f["angs"][i], f["snrs"][i])
                       for i in range(len(f["ages"]))), 2, 2)
    assert np.array_equal(np.stack(res), f["res"], equal_nan=True)


def test_search_grids():
    assert len(orc.angle_grid()) == 181            # core.py:173-175
    assert len(orc.angle_grid(-np.pi / 4, np.pi / 4)) == 91
    assert len(orc.angle_grid(-17 * np.pi / 180, 17 * np.pi / 180)) == 35
    ages = orc.age_grid()
    assert len(ages) == 35 and ages[0] == 1.0 and np.isclose(ages[-1], 10 ** 3.4)


def test_small_searches_reference():
    for c in load_cases("ref_match_small.npz"):
        kw = dict(zip([str(k) for k in c["keys"]], [float(v) for v in c["vals"]]))
        kind = str(c["kind"])
        if kind == orc.RICKER:
            continue        # even template: ties decided by FFT rounding, see test below
        res = orc.match(c["z"], float(c["dx"]), float(c["dy"]), kind, **kw)
        assert np.allclose(np.stack(res), c["res"], rtol=1e-7, atol=1e-10)


def test_small_search_ricker_near_tie():
    for c in load_cases("ref_match_small.npz"):
        if str(c["kind"]) != orc.RICKER:
            continue
        kw = dict(zip([str(k) for k in c["keys"]], [float(v) for v in c["vals"]]))
        angs = orc.angle_grid(kw["ang_min"], kw["ang_max"])
        a_st, s_st = orc.snr_stack(c["z"], float(c["dx"]), float(c["dy"]), orc.RICKER,
                                   kw["scale"], [kw["age"]], angs)
        chk = orc.check_fold(c["res"], a_st[0], s_st[0], np.full(len(angs), kw["age"]), angs,
                             tie_rtol=1e-9, amp_tol=(1e-9, 1e-12), snr_tol=(1e-7, 1e-10))
        assert chk["n_bad"] == 0 and chk["n_strict"] > 0.9 * chk["n"]


def test_synthetic_match_single_age_golden():
    # scarplet/tests/test_core.py:47-64 (synthetic_match2.npy)
    z = np.load(golden("ref_synthetic_dem.npy"))
    gold = np.load(golden("ref_synthetic_match2.npz"))["res"]
    res = orc.match(z, 1.0, 1.0, orc.SCARP, scale=100, age=10,
                    ang_max=np.pi / 2, ang_min=-np.pi / 2)
    assert np.allclose(res, gold, rtol=1e-7, atol=1e-10)


@pytest.mark.slow
def test_synthetic_match_full_grid_golden():
    # scarplet/tests/test_core.py:28-45 (synthetic_match1.npy): 35 x 181 templates
    z = np.load(golden("ref_synthetic_dem.npy"))
    gold = np.load(golden("ref_synthetic_match1.npz"))["res"]
    res = orc.match(z, 1.0, 1.0, orc.SCARP, scale=100,
                    ang_max=np.pi / 2, ang_min=-np.pi / 2)
    assert np.allclose(np.stack(res), gold, rtol=1e-5, atol=1e-8)


def test_real_space_closed_form():
    rng = np.random.default_rng(0)
    for ny, nx in [(33, 36), (34, 35)]:
        z = rng.standard_normal((ny, nx))
        curv = orc.directional_curvature(z, 1.0, 1.0, 0.4)
        W = orc.scarp_template(6, 2.0, 0.4, nx, ny, 1.0)
        _, _, det = orc.match_arrays(curv, W, np.zeros((ny, nx), bool), details=True)
        assert np.allclose(orc.xcorr_direct(curv, W), det["xcorr"], atol=1e-12)
        assert np.allclose(orc.xcorr_direct(curv ** 2, (W != 0).astype(float)), det["T3"], atol=1e-12)


def test_window_limit_axes_is_the_mask():
    # the separable restatement equals WindowedTemplate.py:66-84 cell for cell
    for (nx, ny, de, alpha, c, d) in [(60, 50, 1.0, 0.3, 4.0, 9.0), (61, 47, 2.0, -np.pi / 4, 7.0, 20.0),
                                      (40, 40, 1.0, np.pi / 2, 2.0, 6.0)]:
        xm, ym = orc.window_limit_axes(nx, ny, de, alpha, c, d)
        assert np.array_equal(ym[:, None] | xm[None, :], orc.window_limits(nx, ny, de, alpha, c, d))


@pytest.mark.parametrize("shape", [(170, 160), (171, 161)])
def test_window_stack_equals_whole_dem_stack(shape):
    """snr_stack_window (used at 10000 x 10000, where whole-DEM templates are
    out of reach) against snr_stack on DEMs small enough for both: interior,
    border and wrap-corner windows, with the full grid's masks."""
    ny, nx = shape
    o = ny % 2
    rng = np.random.default_rng(3)
    z = np.cumsum(rng.standard_normal((ny, nx)), 1) * 0.05 + rng.standard_normal((ny, nx)) * 0.03
    angles = [-np.pi / 4, 0.3, np.pi / 2]
    for kind, scale, pars, margin in ((orc.SCARP, 8, [1.0, 10.0], 40), (orc.RICKER, 5, [0.2], 60),
                                      (orc.RIGHT_UPPER, 8, [4.0], 40)):
        a, s = orc.snr_stack(z, 1.0, 1.0, kind, scale, pars, angles)
        for win in [(0, 20 + o, 0, 24 + o), (ny - 20 - o, ny, nx - 22 - o, nx), (60, 90 + o, 50, 80 + o)]:
            aw, sw = orc.snr_stack_window(z, 1.0, 1.0, kind, scale, pars, angles, win, margin)
            i0, i1, j0, j1 = win
            assert np.allclose(aw, a[:, :, i0:i1, j0:j1], rtol=1e-9, atol=1e-12 * np.abs(a).max())
            assert np.allclose(sw, s[:, :, i0:i1, j0:j1], rtol=1e-7, atol=1e-10 * s.max())
            assert np.array_equal(sw == 0, s[:, :, i0:i1, j0:j1] == 0)


def test_windows_direct_equals_windows_by_fft():
    """snr_stack_windows_direct (real-space closed form at a few cells: what the GPU tests probe the exact mode's cells
    with) against snr_stack_window (the reference's FFT form on the crop): every template family, even and odd sizes,
    windows on the DEM's wrap edges.  Float64 summation order apart (1e-12), the same numbers and the same masks."""
    rng = np.random.default_rng(1)
    for (ny, nx, de, kind, scale, ages, m) in ((120, 116, 1.0, orc.SCARP, 8., [1., 10.], 30), (121, 116, 2.0, orc.SCARP, 12., [2.], 30),
                                               (120, 117, 1.0, orc.RICKER, 3., [0.3], 50), (117, 119, 1.0, orc.RIGHT_UPPER, 6., [5.], 30),
                                               (116, 116, 1.0, orc.LEFT_UPPER, 6., [5.], 30)):
        z = np.cumsum(rng.standard_normal((ny, nx)), 0) * 0.03 + rng.standard_normal((ny, nx)) * 0.05
        angles = orc.angle_grid()[::45]
        h, w = 2 + ny % 2, 2 + nx % 2
        wins = [(5, 5 + h, 7, 7 + w), (ny - h, ny, nx - w, nx), (0, h, 0, w), (60, 60 + h, 40, 40 + w)]
        dy = -de if kind == orc.RICKER else de
        st = orc.snr_stack_windows_direct(z, de, dy, kind, scale, ages, angles, wins, m)
        for wn, (a, s) in zip(wins, st):
            a2, s2 = orc.snr_stack_window(z, de, dy, kind, scale, ages, angles, wn, m)
            assert ((a2 == 0) == (a == 0)).all() and ((s2 == 0) == (s == 0)).all(), kind
            assert np.all(np.abs(a - a2) <= 1e-10 * (np.abs(a2) + 1e-9 * np.max(np.abs(a2)))), kind
            assert np.all(np.abs(s - s2) <= 1e-10 * (np.abs(s2) + 1e-9 * np.max(s2))), kind
