"""BASELINE.json's configurations other than the bench workload, as parity
cases on the device (C1 Carrizo lidar, C2 2048^2 synthetic, C5 Grand Canyon
channels), plus the batch chunking of one orientation run.

DEM fixtures: tests/golden/dem_carrizo.npz and dem_grandcanyon.npz hold the
rasters of the reference's sample datasets (scarplet/datasets/data/*.tif)
converted to arrays; tolerances and the near-tie policy are those of
test_gpu_parity.py.
"""
import numpy as np
import pytest

import scarplet_oracle as orc
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic
from scarplet_amd import WindowedTemplate as WT
from conftest import golden
from test_gpu_parity import (AMP_RTOL, AMP_ATOL, SNR_RTOL, SNR_ATOL, TIE_RTOL, EXACT_MIN,
                             fold_check, grid, report, CLS)

pytestmark = pytest.mark.gpu


def dem_fixture(name):
    f = np.load(golden(name))
    return f["z"].astype(float), float(f["dx"]), float(f["dy"])


def test_c1_carrizo_single_age_35_orientations(gpu_ctx):
    """configs[0]: load_carrizo(), Scarp, scale=100, age=10, 35 orientations
    (one-degree steps over +-17 degrees), 900 x 505 lidar DEM at 2 m."""
    z, dx, dy = dem_fixture("dem_carrizo.npz")
    assert z.shape == (900, 505) and (dx, dy) == (2.0, 2.0)
    lim = 17 * np.pi / 180
    angles = _plan.angle_grid(-lim, lim)
    assert len(angles) == 35
    res = sl.match(grid(z, dx, dy), sl.Scarp, scale=100, age=10, ang_min=-lim, ang_max=lim)
    assert res.shape == (4,) + z.shape
    chk = fold_check(res, z, dx, dy, orc.SCARP, 100, [10.0], angles)
    report("C1 carrizo 1 x 35", chk, "auto")
    assert chk["n_bad"] == 0, chk
    assert chk["exact_frac"] >= EXACT_MIN, chk
    assert set(np.unique(res[1])) <= {0.0, 10.0}


def test_c5_grandcanyon_channel_readme_example(gpu_ctx):
    """configs[4], single width: the README / channels.ipynb call
    sl.match(load_grandcanyon(), Channel, scale=10, age=0.1, +-pi/2) with
    dx = 1, dy = -1 as the notebook sets them."""
    z, dx, dy = dem_fixture("dem_grandcanyon.npz")
    assert z.shape == (512, 512)
    res = sl.match(grid(z, dx, dy), sl.Channel, scale=10., age=0.1,
                   ang_min=-np.pi / 2, ang_max=np.pi / 2)
    chk = fold_check(res, z, dx, dy, orc.RICKER, 10., [0.1], _plan.angle_grid())
    # (measured: 11 of 262 144 cells carry the runner-up, gap <= 7e-4: the float32 FFT convolution's SNR
    #  error on this int16 DEM is 2e-4; the real-space path is exact on it, "C5 scale ... direct" below)
    report("C5 grand canyon channel 1 x 181", chk, "auto", max_inexact=16)
    assert chk["n_bad"] == 0, chk
    # exact=True: the cells where the FFT row pass saw a near-tie are searched again on the real-space path.  A
    # Ricker wavelet's SNR varies slowly with the orientation: a QUARTER of this DEM's cells hold a second template
    # within the window (68 000 flagged of 262 144) - the mode then answers with the real-space search of the whole
    # DEM (measured alone: 5 cells whose two best templates lie 1e-6 apart in the oracle's float64 SNRs, against 10
    # cells up to 7e-5 apart on the FFT path), and the cells THAT path decides inside its own rounding (~3 000) are
    # scored in float64 for all 181 templates (sc_score_cells_f64): every decidable cell carries the oracle's argmax
    # Round 5, end: the row pass LISTS its near-ties (sc_get_near_events) and only the (cell, template) pairs the list names
    # are scored in float64 (sc_score_pairs_f64) - no second search; the longer route stays for lists that overflow.
    for events in (False, True):
        m = sl.Matcher(grid(z, dx, dy), ctx=gpu_ctx)
        m.EXACT_USE_EVENTS = events
        if events:
            res = m.search(sl.Channel, 10., [0.1], _plan.angle_grid(), method="fft", exact=True).result()
        else:
            with pytest.warns(UserWarning, match="searching the whole DEM on the real-space path"):
                res = m.search(sl.Channel, 10., [0.1], _plan.angle_grid(), method="fft", exact=True).result()
        path = "fft" if events else "direct"                    # (the cells the mode did not touch keep that path's values)
        chk = fold_check(res, z, dx, dy, orc.RICKER, 10., [0.1], _plan.angle_grid(), path)
        print("     exact=True:", m.exact_stats, m.method_used)
        report("C5 grand canyon channel 1 x 181, exact=True%s" % (" (events)" if events else ""), chk, path, max_inexact=0)
        assert chk["n_bad"] == 0, chk
        assert m.method_used == path and m.exact_stats["flagged_cells"] > 0.03 * z.size and m.exact_stats["float64_cells"] > 0
        assert (m.exact_stats.get("route") == "device") == events, m.exact_stats
    res2 = sl.match(grid(z, dx, dy), sl.Channel, scale=10., age=0.1, ang_min=-np.pi / 2, ang_max=np.pi / 2, exact=True)
    # (the public keyword reaches the same path: the same (age, orientation) in every cell; the float64-scored cells'
    #  amplitudes and SNRs agree to 1e-12 - the templates' sum(W**2) is accumulated with float64 atomics, whose order
    #  is the last bits of those values)
    assert np.array_equal(np.stack(res)[1:3], res2[1:3])
    assert np.allclose(np.stack(res)[[0, 3]], res2[[0, 3]], rtol=1e-12, atol=0)


def test_c5_grandcanyon_channel_five_scales(gpu_ctx):
    """configs[4] as SURVEY.md section 8(d) defines it: Channel (Ricker) plugin, f = 0.1,
    scales {5, 10, 20, 40, 80}, one result set per scale (every sixth degree here so
    the oracle stacks stay small; the full 181-orientation grid is the test above)."""
    z, dx, dy = dem_fixture("dem_grandcanyon.npz")
    angles = _plan.angle_grid()[::6]
    for scale in (5., 10., 20., 40., 80.):
        a_st, s_st = orc.snr_stack(z, dx, dy, orc.RICKER, scale, [0.1], angles, workers=4)
        T = len(angles)
        for method in ("fft", "direct") if scale <= 10. else ("fft",):
            m = sl.Matcher(grid(z, dx, dy), ctx=gpu_ctx)
            res = m.search(WT.Channel, scale, [0.1], angles, method=method).result()
            chk = orc.check_fold(res, a_st.reshape(T, *z.shape), s_st.reshape(T, *z.shape),
                                 np.repeat([0.1], T), angles,
                                 tie_rtol=orc.tie_window(method, orc.RICKER), amp_tol=(AMP_RTOL, AMP_ATOL * np.max(np.abs(a_st))),
                                 snr_tol=(orc.snr_tolerance(orc.RICKER)[0], SNR_ATOL * np.max(s_st)))
            report("C5 scale %g %s" % (scale, method), chk, method)
            assert chk["n_bad"] == 0, (scale, method, chk["n_bad"])


def test_c5_match_scales_equals_one_match_per_scale(gpu_ctx):
    """sl.match_scales (round 6: BASELINE config C5 through ONE entry point - the DEM uploaded once, the orientations'
    curvature spectra computed for the first scale and kept for the others) against one sl.match per scale, as the reference
    runs the job (channels.ipynb): the same bits in every plane, in the default (exact) and in the float32 mode."""
    z, dx, dy = dem_fixture("dem_grandcanyon.npz")
    scales = [5., 10., 20., 40., 80.]
    for exact in (None, False):
        many = sl.match_scales(grid(z, dx, dy), sl.Channel, scales, age=0.1, exact=exact)
        assert len(many) == 5 and all(r.shape == (4,) + z.shape for r in many)
        for sc, r in zip(scales, many):
            one = sl.match(grid(z, dx, dy), sl.Channel, scale=sc, age=0.1, exact=exact)
            assert np.array_equal(r, one), (sc, exact)
    assert not np.array_equal(many[0][3], many[1][3])


def test_channel_several_widths_in_one_fold(gpu_ctx):
    """The second constructor argument of Ricker is the wavelet frequency: several of
    them fold in one device search like ages do for Scarp."""
    z, dx, dy = dem_fixture("dem_grandcanyon.npz")
    widths = [0.05, 0.1, 0.2, 0.4, 0.8]
    angles = _plan.angle_grid()[::6]
    a_st, s_st = orc.snr_stack(z, dx, dy, orc.RICKER, 10., widths, angles, workers=4)
    T = len(widths) * len(angles)
    m = sl.Matcher(grid(z, dx, dy), ctx=gpu_ctx)
    res = m.search(WT.Channel, 10., widths, angles, method="fft").result()
    chk = orc.check_fold(res, a_st.reshape(T, *z.shape), s_st.reshape(T, *z.shape),
                         np.repeat(widths, len(angles)), np.tile(angles, len(widths)),
                         tie_rtol=TIE_RTOL, amp_tol=(AMP_RTOL, AMP_ATOL * np.max(np.abs(a_st))),
                         snr_tol=(orc.snr_tolerance(orc.RICKER)[0], SNR_ATOL * np.max(s_st)))
    assert chk["n_bad"] == 0, chk["n_bad"]
    assert len(np.unique(res[1][res[3] > 0])) > 1     # more than one width wins somewhere


def test_c2_synthetic_2048_ten_ages_91_orientations(gpu_ctx):
    """configs[1]: 2048 x 2048 synthetic DEM (circular FFT mode, T = 2048),
    Scarp, 10 ages x 91 orientations.  The oracle cannot build the 910-map
    stack at this size, so a sample of templates is checked instead:
      * max property: the folded SNR is >= every sampled template's SNR;
      * where a sampled template won, amp / SNR equal the oracle's map;
      * every winner lies on the searched grid."""
    n = 2048
    g = synthetic.synthetic_scarp(n)
    z = g._griddata
    ages = _plan.age_grid()[np.round(np.linspace(0, 34, 10)).astype(int)]
    angles = _plan.angle_grid(-np.pi / 4, np.pi / 4)
    assert len(ages) == 10 and len(angles) == 91
    m = sl.Matcher(g, ctx=gpu_ctx)
    p = m.search(sl.Scarp, 100, ages, angles, method="fft")
    amp, age, ang, snr = m.result()
    assert np.isin(age[snr > 0], ages).all() and np.isin(ang[snr > 0], angles).all()

    won = 0
    for ia, ib in [(0, 0), (3, 45), (4, 45), (9, 90), (7, 20), (2, 70)]:
        o_amp, _, _, o_snr = orc.match_template(z, 1.0, 1.0, orc.SCARP, 100, ages[ia], angles[ib], workers=4)
        tol_s = SNR_RTOL * o_snr + SNR_ATOL * np.max(o_snr)
        assert (snr >= o_snr * (1 - TIE_RTOL) - tol_s).all(), (ia, ib)
        mine = np.isclose(age, ages[ia], rtol=1e-9) & (ang == angles[ib])
        won += int(mine.sum())
        assert (np.abs(snr - o_snr)[mine] <= tol_s[mine]).all(), (ia, ib)
        tol_a = AMP_RTOL * np.abs(o_amp) + AMP_ATOL * np.max(np.abs(o_amp))
        assert (np.abs(amp - o_amp)[mine] <= tol_a[mine]).all(), (ia, ib)
    assert won > 0


def test_c2_windows_against_all_910_templates(gpu_ctx, oracle_pool):
    """configs[1] in full: windows of the folded 2048 x 2048 result against EVERY one of the
    10 x 91 templates (oracle.snr_stack_window builds the 910-map stack of a window from the
    periodic DEM, as bench.py's own check does) - inside the DEM, on the wrap corner and on a
    wrap edge; exact argmax, values and the window policy asserted."""
    n = 2048
    g = synthetic.synthetic_scarp(n)
    ages = _plan.age_grid()[np.round(np.linspace(0, 34, 10)).astype(int)]
    angles = _plan.angle_grid(-np.pi / 4, np.pi / 4)
    m = sl.Matcher(g, ctx=gpu_ctx)
    m.search(sl.Scarp, 100, ages, angles, method="fft")
    res = m.result()
    T = len(ages) * len(angles)
    assert T == 910
    for name, win in (("interior", (700, 748, 1300, 1348)), ("wrap corner", (0, 48, n - 48, n)),
                      ("top wrap edge", (n - 48, n, 900, 948))):
        i0, i1, j0, j1 = win
        a_st, s_st = orc.snr_stack_window(g._griddata, 1.0, 1.0, orc.SCARP, 100, ages, angles, win, 160,
                                          pool=oracle_pool)
        sub = tuple(np.asarray(r)[i0:i1, j0:j1] for r in res)
        chk = orc.check_fold(sub, a_st.reshape(T, i1 - i0, j1 - j0), s_st.reshape(T, i1 - i0, j1 - j0),
                             np.repeat(ages, len(angles)), np.tile(angles, len(ages)),
                             tie_rtol=orc.tie_window("fft", orc.SCARP), amp_tol=(AMP_RTOL, AMP_ATOL * np.max(np.abs(a_st))),
                             snr_tol=(SNR_RTOL, SNR_ATOL * np.max(s_st)))
        report("C2 all 910 templates, window %s" % name, chk, "fft")
        assert chk["n_bad"] == 0, (name, chk["n_bad"])
        assert chk["exact_frac"] >= EXACT_MIN, (name, chk["exact_frac"])


def test_more_parameters_than_one_batch(gpu_ctx):
    """An orientation run longer than the device batch (64 templates) is
    split into chunks that fold into the same running best."""
    rng = np.random.default_rng(64)
    z = (np.cumsum(np.cumsum(rng.standard_normal((90, 100)), 0), 1) * 0.01
         + rng.standard_normal((90, 100)) * 0.05).astype(np.float32)
    ages = list(10 ** np.linspace(0, 2, 70))
    angles = np.array([-0.4, 0.0, 0.9])
    for method in ("fft", "direct"):
        m = sl.Matcher(grid(z, 1.0), ctx=gpu_ctx)
        res = m.search(WT.Scarp, 8, ages, angles, method=method).result()
        chk = fold_check(res, z, 1.0, 1.0, orc.SCARP, 8, ages, angles, method)
        report("70 ages (two batches) %s" % method, chk, method)
        assert chk["n_bad"] == 0, (method, chk)
        assert chk["exact_frac"] >= EXACT_MIN, chk


def test_odd_tile_count_and_odd_template_count(gpu_ctx):
    """A tile pair whose second tile is empty carries two TEMPLATES per transform
    instead (k_inv_cols_sym / k_inv_rows_fast, "paired templates"): exercised with an
    odd number of tiles and an odd number of templates per orientation."""
    ages, angles = [2.0, 20.0, 200.0], np.array([-0.8, 0.3])
    for n in range(600, 1000, 20):
        g = synthetic.synthetic_scarp(n, ny=n - 30, seed=n)
        m = sl.Matcher(g, ctx=gpu_ctx)
        arr, bbox, area = m.describe(WT.Scarp, 30, ages, angles)
        p = _plan.Plan(m.ny, m.nx, m.core, bbox, whole=True, method=_plan.METHOD_FFT, t_max=512)
        if (p.nty * p.ntx) % 2 == 1 and p.nty * p.ntx > 1:
            break
    else:
        pytest.fail("no DEM size with an odd tile count found")
    sp = sl._lib.sc_plan(method=1, Ty=p.Ty, Tx=p.Tx, Vy=p.Vy, Vx=p.Vx, nty=p.nty, ntx=p.ntx,
                         circ_y=int(p.circ_y), circ_x=int(p.circ_x), Py=p.Py, Qx=p.Qx, group=len(ages))
    for exact in (False, True):
        # (exact: the row pass flags near-ties - option "near_window", what Matcher.search(exact=True) sets - and the
        #  flagged cells are searched again on the real-space path; by hand here because the plan is)
        m.ctx.reset_best()
        m._patches, m._cells64, m.plan, m.method_used = [], None, p, "fft"
        m.ctx.set_option("near_window", m.EXACT_WINDOW[WT.KIND_SCARP] if exact else 0.0)
        m.ctx.match(arr, sp, sync=True)
        m.ctx.set_option("near_window", 0.0)
        m.params, m.angles, m.n_templates = np.asarray(ages), angles, len(arr)
        m._id_par, m._id_ang = np.repeat(ages, len(angles)), np.tile(angles, len(ages))
        if exact:
            m._rescore_near_ties(WT.Scarp, 30, np.asarray(ages), angles, {})
            print("     exact:", m.exact_stats)
        res = m.result()
        chk = fold_check(res, g._griddata, 1.0, 1.0, orc.SCARP, 30, ages, angles)
        # (without the exact mode: 1 of 748 000 cells measured)
        report("odd tile count, paired templates%s" % (", exact" if exact else ""), chk, max_inexact=0 if exact else 2)
        assert chk["n_bad"] == 0, (p, chk)
        assert chk["exact_frac"] >= EXACT_MIN, chk
    m._id_par = None


def test_paired_templates_agree_with_paired_tiles():
    """Device self-consistency on a DEM too large for the oracle stack: the
    paired-template mode of an unpaired tile (default) against the same search
    with that tile in a half-empty tile pair (option variant=5 switches the mode off)."""
    g = synthetic.synthetic_scarp(1700, ny=1650, seed=11)
    # (+pi/2 left out: Scarp at -pi/2 and +pi/2 is one template up to the sign of W, their
    #  SNRs tie to an ulp and the two modes may keep either)
    ages, angles = _plan.age_grid()[::5], _plan.angle_grid()[:-1:30]
    out = {}
    for variant in ("0", "5"):
        ctx = sl._lib.Context(0)
        ctx.set_option("variant", int(variant))
        m = sl.Matcher(g, ctx=ctx)
        out[variant] = m.search(sl.Scarp, 100, ages, angles, method="fft").result()
        assert (m.plan.nty * m.plan.ntx) % 2 == 1, m.plan
        del m
        ctx.close()
    a, b = out["0"], out["5"]
    same = (a[1] == b[1]) & (a[2] == b[2])
    assert same.mean() > 0.999, float(same.mean())
    assert np.allclose(a[3][same], b[3][same], rtol=2e-4, atol=1e-7 * b[3].max())
    assert np.allclose(a[0][same], b[0][same], rtol=2e-4, atol=1e-7 * np.abs(b[0]).max())
    # where the winner differs the two SNRs are a near-tie
    assert np.allclose(a[3][~same], b[3][~same], rtol=TIE_RTOL)


def test_symmetric_spectrum_path_agrees_with_complex_path():
    """The real-coefficient path of Scarp / Ricker templates (k_split_templ_sym,
    k_inv_cols_sym) against the complex-spectrum path every other template takes
    (option variant=8 forces it), on several tiles of an even x odd DEM."""
    g = synthetic.synthetic_scarp(2600, ny=2301, seed=12)
    ages, angles = _plan.age_grid()[2::8], _plan.angle_grid()[7::40]
    out = {}
    for variant in ("0", "8"):
        ctx = sl._lib.Context(0)
        ctx.set_option("variant", int(variant))
        m = sl.Matcher(g, ctx=ctx)
        out[variant] = m.search(sl.Scarp, 100, ages, angles, method="fft").result()
        assert m.plan.nty * m.plan.ntx > 1
        del m
        ctx.close()
    a, b = out["0"], out["8"]
    same = (a[1] == b[1]) & (a[2] == b[2])
    assert same.mean() > 0.999, float(same.mean())
    assert np.allclose(a[3][same], b[3][same], rtol=2e-4, atol=1e-7 * b[3].max())
    assert np.allclose(a[0][same], b[0][same], rtol=2e-4, atol=1e-7 * np.abs(b[0]).max())
    assert np.allclose(a[3][~same], b[3][~same], rtol=TIE_RTOL)


def test_orientation_batching_is_bit_identical():
    """Small searches send several orientations through every launch (sc_fft.hip "Orientation
    batching"): same cells, same fold order - the record must equal the one-orientation-per-
    launch path bit for bit.  Covers one- and two-tile plans, paired-template tiles, single and
    several ages, the masked (UpperBreak) and the generic complex-spectrum (Shifted) paths."""
    rng = np.random.default_rng(77)

    def dem(ny, nx):
        return (np.cumsum(np.cumsum(rng.standard_normal((ny, nx)), 0), 1) * 0.01
                + rng.standard_normal((ny, nx)) * 0.05).astype(np.float32)
    gc = dem_fixture("dem_grandcanyon.npz")
    cases = [
        (grid(dem(900, 505), 2.0), WT.Scarp, 100, [10.0], _plan.angle_grid(-0.3, 0.3), {}),          # C1 shape: 2 tiles
        (grid(gc[0], gc[1], gc[2]), WT.Channel, 10., [0.1], _plan.angle_grid()[::3], {}),             # C5: one tile, one template
        (grid(gc[0], gc[1], gc[2]), WT.Channel, 20., [0.05, 0.1, 0.2], _plan.angle_grid()[::7], {}),  # odd template count
        (synthetic.synthetic_scarp(1024, seed=5), WT.Scarp, 60, _plan.age_grid()[::4], _plan.angle_grid()[::9], {}),
        (grid(dem(600, 700), 1.0), WT.RightFacingUpperBreakScarp, 20, [3.0, 30.0], _plan.angle_grid()[::10], {}),
        (grid(dem(520, 530), 1.0), WT.ShiftedLeftFacingUpperBreakScarp, 12, [4.0], _plan.angle_grid()[::12],
         dict(dx=3, dy=-2)),
    ]
    for ic, (g, cls, scale, params, angles, kw) in enumerate(cases):
        out = []
        # case 1 (one tile, one template per orientation) batches its orientations in PAIRS
        # (inv_cols_sym_body, XP): two orientations share a transform, so the record equals the
        # unbatched one only to rounding; with that form off (variant 12) it is equal in every bit
        xp_case = ic == 1
        for batch, variant in ((1, 12 if xp_case else 0), (0, 0)) + (((1, 0),) if xp_case else ()):
            ctx = sl._lib.Context(0)
            ctx.set_option("batch", batch)
            ctx.set_option("variant", variant)
            m = sl.Matcher(g, ctx=ctx)
            m.search(cls, scale, params, angles, method="fft", **kw)
            out.append(m.ctx.get_best())
            plan = m.plan
            m.ctx.clear_windows()
            del m
            ctx.close()
        for a, b, name in zip(out[0], out[1], ("amp", "snr", "id")):
            assert np.array_equal(a, b), (cls.__name__, str(plan), name, int((a != b).sum()))
        assert (out[0][1] > 0).any()
        if xp_case:
            amp, snr, idx = out[2]
            same = idx == out[1][2]
            # (a Channel is the same template at -pi/2 and +pi/2: the cells those two win are exact ties
            #  in exact arithmetic, and which of them keeps the cell is a matter of rounding)
            assert same.mean() >= 0.98, same.mean()
            assert not np.array_equal(snr, out[1][1])                      # (the paired form did run)
            scale_s = float(out[1][1].max())
            assert np.abs(snr - out[1][1])[same].max() <= 2e-5 * scale_s
            assert np.abs(amp - out[1][0])[same].max() <= 2e-5 * float(np.abs(out[1][0]).max())
            # where the orientation differs the two candidates tie within the FFT path's window
            if (~same).any():
                assert (np.abs(snr - out[1][1])[~same] <= TIE_RTOL * np.maximum(snr, out[1][1])[~same]).all()


def test_fused_template_split_is_bit_identical():
    """k_fwd_cols_tsym (column transform + split of symmetric templates in one kernel) against
    the two-kernel sequence it replaces (option variant=7): identical coefficients, so an
    identical record; on an even x odd DEM and on a circular single tile."""
    for g, scale, ages in ((synthetic.synthetic_scarp(1300, ny=1201, seed=21), 60, _plan.age_grid()[3::9]),
                           (synthetic.synthetic_scarp(1024, seed=22), 40, _plan.age_grid()[::6])):
        out = []
        for variant in (0, 7):
            ctx = sl._lib.Context(0)
            ctx.set_option("variant", variant)
            m = sl.Matcher(g, ctx=ctx)
            m.search(sl.Scarp, scale, ages, _plan.angle_grid()[::16], method="fft")
            out.append(m.ctx.get_best())
            del m
            ctx.close()
        for a, b, name in zip(out[0], out[1], ("amp", "snr", "id")):
            assert np.array_equal(a, b), (name, int((a != b).sum()))
    gc = dem_fixture("dem_grandcanyon.npz")
    out = []
    for variant in (0, 7):
        ctx = sl._lib.Context(0)
        ctx.set_option("variant", variant)
        m = sl.Matcher(grid(gc[0], gc[1], gc[2]), ctx=ctx)
        m.search(WT.Channel, 10., [0.1, 0.2], _plan.angle_grid()[::9], method="fft")
        out.append(m.ctx.get_best())
        del m
        ctx.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_column_pass_forms_are_bit_identical():
    """The three forms of the inverse column pass: k_inv_cols_w8 (default at column lengths 1024 and
    2048: one wave per column), k_inv_cols_symx (variant=2: four columns per workgroup, block and
    mirror workgroups in one launch, paired per XCD) and the two-launch pass (variant=6)
    compute the same Y cell for cell, so the same record - at T = 2048 with several
    tile pairs and a partial last tile row, at T = 1024 / 512, on non-square tiles, with an odd
    tile count (paired-template mode) and with batched orientations on a circular tile.  Round 4:
    the wave-per-column kernel does not store, and the row pass does not transform, the rows a
    template's window limits mask (variant=13 switches both off): the record must not notice.  Nor
    must it notice how many tile pairs share a launch of the wave-per-column pass (option i1_pairs)."""
    a9 = _plan.age_grid()[::4]                          # 9 ages
    cases = [(synthetic.synthetic_scarp(3900, ny=3700, seed=31), sl.Scarp, 100, a9, _plan.angle_grid()[3::45]),
             (synthetic.synthetic_scarp(1500, ny=1400, seed=32), sl.Scarp, 40, a9, _plan.angle_grid()[::30]),
             (synthetic.synthetic_scarp(700, seed=33), sl.Scarp, 12, list(10 ** np.linspace(0, 1.6, 8)), _plan.angle_grid()[::20]),
             # non-square tiles: 1024 x 512, 512 x 1024, 2048 x 512 (batched orientations where they fit)
             (synthetic.synthetic_scarp(505, ny=900, seed=35), sl.Scarp, 50, a9, _plan.angle_grid()[::18]),
             (synthetic.synthetic_scarp(900, ny=505, seed=36), sl.Scarp, 50, a9, _plan.angle_grid()[::18]),
             (synthetic.synthetic_scarp(420, ny=3000, seed=37), sl.Scarp, 60, a9, _plan.angle_grid()[::25]),
             (synthetic.synthetic_scarp(1024, seed=34), sl.Channel, 15, [0.05, 0.07, 0.1, 0.14, 0.2, 0.28, 0.4, 0.56], _plan.angle_grid()[::12])]
    for ci, (g, cls, scale, params, angles) in enumerate(cases):
        out = []
        # (i1_pairs: tile pairs per launch of the wave-per-column pass, interleaved along x - default 2; one pair per
        #  launch and three - a chunk's last launch then holds fewer - must give the same record)
        runs = [(0, None), (2, None), (6, None), (13, None)] + ([(0, 1), (0, 3)] if ci < 2 else [])
        for variant, pairs in runs:                      # 13: no skipping of the rows masked by window limits
            ctx = sl._lib.Context(0)
            ctx.set_option("variant", variant)
            if pairs is not None:
                ctx.set_option("i1_pairs", pairs)
            m = sl.Matcher(g, ctx=ctx)
            m.search(cls, scale, params, angles, method="fft")
            out.append(m.ctx.get_best())
            plan = m.plan
            del m
            ctx.close()
        print("wave-per-column vs paired-launch vs two-launch column pass:", plan)
        for other in out[1:]:
            for a, b, name in zip(out[0], other, ("amp", "snr", "id")):
                assert np.array_equal(a, b), (str(plan), name, int((a != b).sum()))
        assert (out[0][1] > 0).any()


def _fwd_rows_launches(ctx, fn):
    """launches of the curvature/template forward row kernels while fn() runs"""
    ctx.profile(1)
    fn()
    n = ctx.profile_get()["k_fwd_rows"][0]
    ctx.profile(0)
    return n


@pytest.mark.gpu
def test_kept_curvature_spectra_are_reused_and_change_nothing():
    """A search keeps the curvature spectra of its orientations (option "spectra_mb"); the next
    search on the same DEM with the same tiles and orientations - the next scale of a multi-scale
    job, C5 - starts from them.  The record must equal, bit for bit, the one a context without
    kept spectra produces; fewer forward launches show that the spectra were in fact reused; a
    different orientation grid, a different DEM or forget_spectra() recompute them."""
    gc = dem_fixture("dem_grandcanyon.npz")
    g = grid(gc[0], gc[1], gc[2])
    rng = np.random.default_rng(5)
    g2 = grid((gc[0] + 0.05 * rng.standard_normal(gc[0].shape)).astype(gc[0].dtype), gc[1], gc[2])
    angles = _plan.angle_grid()[::4]

    def fresh(gr, scale, ang):
        ctx = sl._lib.Context(0)
        ctx.set_option("spectra_mb", 0)
        m = sl.Matcher(gr, ctx=ctx)
        m.search(WT.Channel, scale, [0.1], ang, method="fft")
        out = m.ctx.get_best()
        ctx.close()
        return out

    ctx = sl._lib.Context(0)
    assert ctx.spectra_mb > 0                       # on by default
    m = sl.Matcher(g, ctx=ctx)
    n_first = _fwd_rows_launches(ctx, lambda: m.search(WT.Channel, 5., [0.1], angles, method="fft"))
    for a, b in zip(m.ctx.get_best(), fresh(g, 5., angles)):
        assert np.array_equal(a, b)
    # the next scale: same orientations -> only the template rows are transformed
    n_second = _fwd_rows_launches(ctx, lambda: m.search(WT.Channel, 20., [0.1], angles, method="fft"))
    assert 0 < n_second < n_first, (n_first, n_second)
    for a, b in zip(m.ctx.get_best(), fresh(g, 20., angles)):
        assert np.array_equal(a, b)
    # the same data through a new Matcher on the same context (one sl.match per scale): not uploaded again
    key = ctx.dem_key
    m2 = sl.Matcher(g, ctx=ctx)
    assert ctx.dem_key == key
    n_third = _fwd_rows_launches(ctx, lambda: m2.search(WT.Channel, 40., [0.1], angles, method="fft"))
    assert n_third == n_second
    for a, b in zip(m2.ctx.get_best(), fresh(g, 40., angles)):
        assert np.array_equal(a, b)
    # other orientations: recomputed
    other = _plan.angle_grid()[1::4]
    n_other = _fwd_rows_launches(ctx, lambda: m2.search(WT.Channel, 40., [0.1], other, method="fft"))
    assert n_other > n_second
    for a, b in zip(m2.ctx.get_best(), fresh(g, 40., other)):
        assert np.array_equal(a, b)
    # forgotten on request, and with another DEM in the context
    ctx.forget_spectra()
    assert _fwd_rows_launches(ctx, lambda: m2.search(WT.Channel, 40., [0.1], other, method="fft")) == n_other
    m3 = sl.Matcher(g2, ctx=ctx)
    assert ctx.dem_key != key
    assert _fwd_rows_launches(ctx, lambda: m3.search(WT.Channel, 40., [0.1], other, method="fft")) == n_other
    for a, b in zip(m3.ctx.get_best(), fresh(g2, 40., other)):
        assert np.array_equal(a, b)
    ctx.close()


def test_device_digest_tells_dems_apart():
    """sc_dem_info: the device's fingerprint of the block it was handed (geometry, cell size, the
    float64 bit patterns), its NaN count, and `unchanged` - the same data handed over again keeps
    the curvature planes; anything else does not."""
    from scarplet_amd import _lib
    ctx = _lib.Context(0)
    rng = np.random.default_rng(0)
    z = rng.standard_normal((40, 50))

    def key(zz, dx=1.0, dy=1.0):
        m = sl.Matcher(sl.DEMGrid.from_array(zz, dx, dy), ctx=ctx)
        return ctx.dem_key, ctx.dem_unchanged, ctx.dem_nan, m
    k0, same, nan, _ = key(z)
    assert not same and nan == 0
    k1, same, nan, _ = key(z.copy())
    assert k1 == k0 and same                              # equal data: recognised on the device
    k2, same, _, _ = key(np.asfortranarray(z))
    assert k2 == k0 and same                              # ... also through a non-contiguous array
    z2 = z.copy(); z2[7, 9] += 1e-12
    k3, same, _, _ = key(z2)
    assert k3 != k0 and not same                          # one bit of one cell
    z3 = z.copy(); z3[[3, 4]] = z3[[4, 3]]
    assert key(z3)[0] != k0                               # the same values elsewhere
    assert key(z, 2.0, 1.0)[0] != k0 and key(z, 1.0, -1.0)[0] != k0
    assert key(z.reshape(50, 40))[0] != k0
    z4 = z.copy(); z4[5, 6] = np.nan; z4[9, 9] = -np.nan
    with pytest.warns(UserWarning, match="NaN"):
        _, same, nan, m = key(z4)
    assert nan == 2 and not same and m.nan_dem
    with pytest.warns(UserWarning, match="NaN"):
        assert not key(z4)[1]                             # a NaN block is never "unchanged"
    assert not key(z)[3].nan_dem
    ctx.close()


@pytest.mark.parametrize("kind,n_ang", [("scarp", 13), ("ricker", 12), ("right_upper_break", 7)])
def test_paired_orientations_against_oracle(gpu_ctx, kind, n_ang):
    """One tile, ONE template per orientation: the batch's orientations ride in pairs through the inverse
    transforms (inv_cols_sym_body, XP) - an odd count leaves the last one alone, an odd template (Scarp)
    takes the factor i, a masked one (UpperBreak) the full epilogue.  Each against the oracle's stack of
    all its templates."""
    rng = np.random.default_rng(404)
    z = (np.cumsum(np.cumsum(rng.standard_normal((512, 512)), 0), 1) * 0.01
         + rng.standard_normal((512, 512)) * 0.05).astype(np.float32).astype(float)
    angles = np.linspace(-1.4, 1.3, n_ang)
    m = sl.Matcher(grid(z, 1.0), ctx=gpu_ctx)
    res = m.search(CLS[kind], 12, [4.0], angles, method="fft").result()
    assert m.plan.circ_y and m.plan.Ty == 512                    # (the plan the paired form applies to)
    chk = fold_check(res, z, 1.0, 1.0, kind, 12, [4.0], angles)
    report("paired orientations %s x %d" % (kind, n_ang), chk, "fft")
    assert chk["n_bad"] == 0, chk
    assert chk["exact_frac"] >= EXACT_MIN, chk


def test_split_row_pass_is_bit_identical():
    """Small grids deal a launch's transforms out over several row workgroups (k_inv_rows_fast SPLITK,
    k_merge_split; option variant=15 switches it off): every share folds in order into a record of its
    own and the shares are merged in order - the record must equal the one-workgroup fold in every bit,
    ties included (a noise-free surface has many)."""
    gc = dem_fixture("dem_grandcanyon.npz")
    cz = dem_fixture("dem_carrizo.npz")
    from scipy.special import erf
    y, x = np.mgrid[-256:256, -256:256].astype(float)
    flat = (-erf((-x * np.sin(1.1) + y * np.cos(1.1)) / (2 * np.sqrt(10.0))) + 0.01 * x).astype(np.float32)
    lim = 17 * np.pi / 180
    cases = [(grid(gc[0], gc[1], gc[2]), sl.Channel, 10.0, [0.1], _plan.angle_grid()),                 # C5: paired orientations
             (grid(gc[0], gc[1], gc[2]), sl.Channel, 20.0, [0.05, 0.1, 0.2], _plan.angle_grid()[::3]),  # paired templates
             (grid(cz[0], cz[1], cz[2]), sl.Scarp, 100.0, [10.0], _plan.angle_grid(-lim, lim)),        # C1: a tile pair, 35 jobs
             (synthetic.synthetic_scarp(700, seed=41), sl.Scarp, 12.0, [1.0, 3.0, 10.0], _plan.angle_grid()[::5]),
             (grid(flat, 1.0), sl.Scarp, 20.0, [3.0, 10.0, 30.0], _plan.angle_grid()[::4])]            # exact ties
    for (g, cls, scale, params, angles) in cases:
        out = []
        for variant in (0, 15):
            ctx = sl._lib.Context(0)
            ctx.set_option("variant", variant)
            m = sl.Matcher(g, ctx=ctx)
            ctx.profile(1)
            m.search(cls, scale, params, angles, method="fft")
            out.append((m.ctx.get_best(), ctx.profile_get()["k_inv_rows"][0], m.plan))
            ctx.close()
        (b0, n0, plan), (b1, n1, _) = out
        print("split row pass:", plan, "row-pass launches", n0, n1)
        for a, b, name in zip(b0, b1, ("amp", "snr", "id")):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (str(plan), name, int((a != b).sum()))
        assert (b0[1] > 0).any()


def test_round5_launch_forms_are_bit_identical():
    """Round 5: (a) the orientation's curvature mixed from the three stencil planes inside the forward row
    pass (option variant=17: the separate k_curv_alpha pass writes the plane and the row pass reads it
    back); (b) an under-filled column pass deals its transforms out along grid.z (split_i1=0: one
    workgroup per column block walks them all); (c) the dealt-out row pass allowed four waves per SIMD
    (split_fill=2048: round 4's two); (d) up to 256 templates per batched launch sequence, the row pass
    folding them in slices of whole orientations of at most 64 templates (batch_templ=64: round 4's 64 per
    sequence; one orientation per sequence, batch=0, is test_orientation_batching_is_bit_identical's); (e) column length
    512 by the half-wave-per-column kernel k_inv_cols_h2 (variant=18: the four-column kernels), with one template, paired
    templates and paired orientations; (f) row-pass launches of up to 255 templates in shares of at most 64 transforms
    (variant=20: at most 64 templates per row-pass launch).  Each is the same arithmetic on the same operands, distributed
    differently: the record must be equal in every bit - ties included - on tiled, paired-template,
    batched and single-template searches."""
    cz = dem_fixture("dem_carrizo.npz")
    gc = dem_fixture("dem_grandcanyon.npz")
    from scipy.special import erf
    y, x = np.mgrid[-150:150, -160:160].astype(float)
    flat = (-erf((-x * np.sin(1.1) + y * np.cos(1.1)) / (2 * np.sqrt(10.0))) + 0.01 * x).astype(np.float32)
    lim = 17 * np.pi / 180
    ages12 = list(_plan.age_grid()[::3])
    cases = [(grid(cz[0], cz[1], cz[2]), sl.Scarp, 100.0, ages12, _plan.angle_grid()[::30]),      # C1F's plan: 3 x 2 tiles of 512, many ages
             (grid(cz[0], cz[1], cz[2]), sl.Scarp, 100.0, [10.0], _plan.angle_grid(-lim, lim)),  # C1: batched orientations
             (grid(gc[0], gc[1], gc[2]), sl.Channel, 10.0, [0.1], _plan.angle_grid()[::2]),      # C5: paired orientations
             (synthetic.synthetic_scarp(1300, seed=5), sl.Scarp, 30.0, ages12[:9], _plan.angle_grid()[::45]),   # 1024 / 2048 tiles
             (grid(flat, 1.0), sl.Scarp, 20.0, [3.0, 10.0, 30.0, 60.0, 90.0, 120.0, 200.0, 300.0], _plan.angle_grid()[::20]),   # exact ties
             (synthetic.synthetic_scarp(700, ny=1100, seed=9), sl.Scarp, 40.0, ages12[:7], _plan.angle_grid()[::25]),   # 512 tiles, odd count: paired templates
             (grid(gc[0], gc[1], gc[2]), sl.Channel, 20.0, [0.05, 0.1, 0.2], _plan.angle_grid()[::9])]                  # one 512 tile, paired templates
    forms = [("default", {}), ("variant 17", {"variant": 17}), ("split_i1 0", {"split_i1": 0}),
             ("split_fill 2048", {"split_fill": 2048}), ("batch_templ 64", {"batch_templ": 64}), ("variant 18", {"variant": 18}), ("variant 20", {"variant": 20}),
             ("all off", {"variant": 17, "split_i1": 0, "split_fill": 2048, "batch_templ": 64}),
             ("variant 18, no split", {"variant": 18, "split_i1": 0})]
    for (g, cls, scale, params, angles) in cases:
        ref = None
        for name, opts in forms:
            ctx = sl._lib.Context(0)
            for k, v in opts.items():
                ctx.set_option(k, v)
            m = sl.Matcher(g, ctx=ctx)
            ctx.profile(1)
            m.search(cls, scale, params, angles, method="fft")
            best, prof, plan = m.ctx.get_best(), ctx.profile_get(), m.plan
            ctx.close()
            if ref is None:
                ref = best
                print("round-5 launch forms:", plan, "launches", {k: v[0] for k, v in prof.items() if v[0]})
                assert (best[1] > 0).any()
                continue
            for a, b, plane in zip(ref, best, ("amp", "snr", "id")):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (str(plan), name, plane, int((a != b).sum()))


def test_c1f_carrizo_full_grid_is_the_reference_flagship(gpu_ctx, oracle_pool):
    """The reference's own flagship call, docs/source/examples/scarps.ipynb cell 12:
    `sl.match(load_carrizo(), Scarp, scale=100.)` - 35 ages x 181 orientations on the 900 x 505 lidar DEM
    ("This can be slow on a laptop!").  Three windows of the folded result - interior, on the seam of four
    FFT tiles, on the wrap corner - against EVERY one of the 6335 templates (oracle.snr_stack_window)."""
    z, dx, dy = dem_fixture("dem_carrizo.npz")
    res = sl.match(grid(z, dx, dy), sl.Scarp, scale=100.)
    assert isinstance(res, tuple) and len(res) == 4 and res[0].shape == z.shape
    ages, angles = _plan.age_grid(), _plan.angle_grid()
    T = len(ages) * len(angles)
    assert T == 6335
    ny, nx = z.shape
    # (crop and DEM sizes must have the same parity: 32 rows of the 900, 33 columns of the 505)
    for name, win in (("interior", (420, 452, 230, 263)), ("tile seam", (332, 364, 240, 273)),
                      ("wrap corner", (0, 32, nx - 33, nx))):
        i0, i1, j0, j1 = win
        a_st, s_st = orc.snr_stack_window(z, dx, dy, orc.SCARP, 100., ages, angles, win, 90, pool=oracle_pool)
        sub = tuple(np.asarray(r)[i0:i1, j0:j1] for r in res)
        chk = orc.check_fold(sub, a_st.reshape(T, i1 - i0, j1 - j0), s_st.reshape(T, i1 - i0, j1 - j0),
                             np.repeat(ages, len(angles)), np.tile(angles, len(ages)),
                             tie_rtol=orc.tie_window("fft", orc.SCARP), amp_tol=(AMP_RTOL, AMP_ATOL * np.max(np.abs(a_st))),
                             snr_tol=(SNR_RTOL, SNR_ATOL * np.max(s_st)))
        report("C1F carrizo 35 x 181, window %s" % name, chk, "auto")
        assert chk["n_bad"] == 0, (name, chk["n_bad"])
