"""exact=True where it ACTS (round 6): the cells whose (age, orientation) the float64 settle changed, and cells it flagged
and left alone, against the oracle's float64 argmax over the WHOLE parameter grid - at the benchmark's size (BASELINE
config C3: 10000 x 10000, 6335 templates), on the reference's flagship call (C1F: carrizo, 6335 templates) and on C2.

The reference's fold is an argmax over float64 SNR maps (compare(), core.py:230-240).  The float32 search decides a cell
to within its own rounding; sc_settle_exact scores the near-tie candidates in float64 and the record takes their argmax.
Round 5 checked the mode on windows with (almost) no near-ties in them; here every probe IS a cell the mode touched:
for each, the oracle builds the (amp, snr) stack of all templates over a small window around it
(oracle.snr_stack_window) and the exact result must carry the oracle's own argmax in every cell of the window
(check_fold's n_inexact == 0: templates whose float64 SNRs agree to 1e-9 - the grid's end twins - are one maximum)."""
import os

import numpy as np
import pytest

import scarplet_oracle as orc
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic
from conftest import golden

pytestmark = pytest.mark.gpu
P = orc.PARITY
N_PROBE = 24


def _fold_twin(ang):
    a = np.array(ang, dtype=float)
    a[np.abs(a - np.pi / 2) < 1e-12] = -np.pi / 2
    return a


def _probe(name, g, cls, kind, scale, ages, angles, gpu_ctx, pool, margin, n_probe=N_PROBE, expect_plan=None):
    z = np.asarray(g._griddata, dtype=float)
    dx, dy = float(g._georef_info.dx), float(g._georef_info.dy)
    ny, nx = z.shape
    m = sl.Matcher(g, ctx=gpu_ctx)
    r0 = np.stack(m.search(cls, scale, ages, angles, method="fft", exact=False).result()).copy()
    r1 = np.stack(m.search(cls, scale, ages, angles, method="fft", exact=True).result())
    st = dict(m.exact_stats)
    assert m.method_used == "fft" and st.get("route") == "device", (m.method_used, st)
    # the same in every bit from run to run: the lists fill in whatever order the atomics take, the scores and the winner
    # do not depend on it (sums in a fixed order, one member of a twin class by rule - sc_settle.hip)
    rec1 = [x.copy() for x in m.ctx.get_best()]
    r1b = np.stack(m.search(cls, scale, ages, angles, method="fft", exact=True).result())
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(rec1, m.ctx.get_best()))
    assert np.array_equal(r1, r1b) and dict(m.exact_stats) == st
    del r1b
    if expect_plan:
        assert (m.plan.Ty, m.plan.nty, m.plan.ntx) == expect_plan, m.plan
    flags = m.ctx.near_ties() != 0
    changed = (r0[1] != r1[1]) | (_fold_twin(r0[2]) != _fold_twin(r1[2]))
    print("%s: exact=True %s; %d cells flagged, %d changed (age, orientation)" % (name, st, int(flags.sum()), int(changed.sum())))
    assert int(flags.sum()) == st["flagged_cells"] and int(changed.sum()) == st["changed_cells"], (st, flags.sum(), changed.sum())
    assert not (changed & ~flags).any()                     # nothing moves outside the flagged cells
    # amp / snr outside the flagged cells: the float32 search's, bit for bit
    assert np.array_equal(r0[:, ~flags], r1[:, ~flags])
    rng = np.random.default_rng(6)
    ch = np.argwhere(changed)
    un = np.argwhere(flags & ~changed)
    ch = ch[rng.permutation(len(ch))[:n_probe]]
    un = un[rng.permutation(len(un))[:n_probe]]
    assert len(un) >= min(n_probe, 1)
    T = len(ages) * len(angles)
    par, ang = np.repeat(ages, len(angles)), np.tile(angles, len(ages))
    h, w = 2 + ny % 2, 2 + nx % 2                           # (the crop keeps the DEM's parity)
    worst_gap, n_checked = 0.0, 0
    win_fft = max(m.EXACT_WINDOW.values()) if kind == orc.RICKER else min(m.EXACT_WINDOW.values())
    probes = [(label, int(i), int(j)) for label, cells in (("changed", ch), ("flagged, unchanged", un)) for (i, j) in cells]
    wins = [(int(min(i, ny - h)), int(min(i, ny - h)) + h, int(min(j, nx - w)), int(min(j, nx - w)) + w) for (_, i, j) in probes]
    # (real space: the template once per (age, angle), every probe's few cells as the closed-form sum - the same numbers
    #  as snr_stack_window's six FFTs per crop to 1e-12, tests/test_oracle.py, at a hundredth of the cost)
    stacks = orc.snr_stack_windows_direct(z, dx, dy, kind, scale, ages, angles, wins, margin, pool)
    if True:
        for (label, i, j), win, (a_st, s_st) in zip(probes, wins, stacks):
            i0, j0 = win[0], win[2]
            a_st, s_st = a_st.reshape(T, h, w), s_st.reshape(T, h, w)
            sub = tuple(r1[k][win[0]:win[1], win[2]:win[3]] for k in range(4))
            chk = orc.check_fold(sub, a_st, s_st, par, ang, tie_rtol=orc.tie_window("fft", kind),
                                 amp_tol=(P["amp"][0], P["amp"][1] * np.max(np.abs(a_st))),
                                 snr_tol=(P["snr"][0], P["snr"][1] * np.max(s_st)))
            # the probe cell itself: the oracle's SNRs of the two answers
            s_cell = s_st[:, i - i0, j - j0]
            t_arg = int(np.argmax(s_cell))

            def t_of(r):
                return int(np.flatnonzero((par == r[1][i, j]) & (ang == r[2][i, j]))[0])
            t0, t1 = t_of(r0), t_of(r1)
            gap = (s_cell[t1] - s_cell[t0]) / s_cell[t_arg]
            print("  %-18s (%5d, %5d): float32 chose (%7.2f, %+.4f), exact (%7.2f, %+.4f), oracle argmax (%7.2f, %+.4f); "
                  "oracle SNR of exact - float32 choice = %+.2e of the maximum; window: inexact=%d bad=%d"
                  % (label, i, j, par[t0], ang[t0], par[t1], ang[t1], par[t_arg], ang[t_arg], gap, chk["n_inexact"], chk["n_bad"]))
            assert chk["n_bad"] == 0 and chk["n_inexact"] == 0, (name, label, (i, j), chk["n_bad"], chk["n_inexact"])
            assert s_cell[t1] >= s_cell[t_arg] * (1.0 - 1e-9), (name, (i, j))
            # the float32 path's choice was inside the window the mode flags in (else the candidate argument fails)
            assert s_cell[t0] >= s_cell[t_arg] * (1.0 - win_fft), (name, (i, j), s_cell[t0], s_cell[t_arg])
            if label == "changed":
                # a change is an improvement in the oracle's own numbers (or a draw at float64 rounding: templates
                # proportional to each other - windows one cell wide at +-pi/2 on a coarse DEM - score the same SNR)
                assert gap > -1e-9, (name, (i, j), gap)
                worst_gap = max(worst_gap, gap)
            n_checked += 1
    print("%s: %d probe windows x %d templates: every cell carries the oracle's argmax; the changed cells' float32 choice "
          "lay up to %.2e below it" % (name, n_checked, T, worst_gap))
    return st, len(ch), len(un)


def test_exact_c3_changed_cells_against_oracle(gpu_ctx, oracle_pool):
    """BASELINE config C3 (the headline): 67 of 1e8 cells change; 24 of them and 24 flagged-but-unchanged cells, each
    against all 6335 templates."""
    g = synthetic.synthetic_scarp(10000)
    # (margin 159: crops of 320 x 320 - the reach is 143 cells - whose FFTs are a third faster than 322's)
    st, n_ch, n_un = _probe("C3", g, sl.Scarp, orc.SCARP, 100.0, _plan.age_grid(), _plan.angle_grid(), gpu_ctx, oracle_pool, 159,
                            expect_plan=(2048, 6, 6))
    assert n_ch >= N_PROBE and n_un >= N_PROBE, (n_ch, n_un, st)


def test_exact_c1f_changed_cells_against_oracle(gpu_ctx, oracle_pool):
    """The reference's flagship call (scarps.ipynb cell 12): load_carrizo(), Scarp, scale=100, 35 x 181."""
    f = np.load(golden("dem_carrizo.npz"))
    g = sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"]))
    st, n_ch, n_un = _probe("C1F", g, sl.Scarp, orc.SCARP, 100.0, _plan.age_grid(), _plan.angle_grid(), gpu_ctx, oracle_pool, 111)   # (crops 224 x 225)
    assert n_ch >= 1 and n_un >= N_PROBE, (n_ch, n_un, st)


def test_exact_c2_changed_cells_against_oracle(gpu_ctx, oracle_pool):
    """BASELINE config C2: 2048 x 2048, 10 ages x 91 orientations (+-pi/4: no end twins)."""
    g = synthetic.synthetic_scarp(2048)
    ages = _plan.age_grid()[np.round(np.linspace(0, 34, 10)).astype(int)]
    angles = _plan.angle_grid(-np.pi / 4, np.pi / 4)
    st, n_ch, n_un = _probe("C2", g, sl.Scarp, orc.SCARP, 100.0, ages, angles, gpu_ctx, oracle_pool, 159)
    assert n_un >= N_PROBE, (n_ch, n_un, st)
