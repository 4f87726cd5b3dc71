"""Host planning (tiles, halos, descriptors) checked end to end on the CPU:
numpy model of the kernels (tests/pipeline_model.py) vs the oracle."""
import numpy as np
import pytest

import scarplet_oracle as orc
import pipeline_model as pm
from scarplet_amd import _plan, WindowedTemplate as WT

rng = np.random.default_rng(3)
CASES = [
    (WT.Scarp, orc.SCARP, 64, 64, 1., 10, [1., 31.6], [0., -1.2, np.pi / 2], 4096, True),
    (WT.Scarp, orc.SCARP, 61, 75, 1., 8, [3.2], [0.4, -0.9], 4096, True),
    (WT.Scarp, orc.SCARP, 65, 64, 2., 20, [10., 100.], [1.0], 4096, True),
    (WT.Scarp, orc.SCARP, 150, 131, 1., 12, [1., 5.], [0.5], 64, True),
    (WT.Scarp, orc.SCARP, 128, 128, 1., 12, [5.], [-0.2], 64, False),
    (WT.Ricker, orc.RICKER, 64, 72, 1., 5, [0.1], [0.8], 4096, True),
    (WT.Channel, orc.RICKER, 63, 60, 1., 8, [0.2], [np.pi / 2], 4096, True),
    (WT.RightFacingUpperBreakScarp, orc.RIGHT_UPPER, 64, 66, 1., 10, [10.], [0.2], 4096, True),
    (WT.LeftFacingUpperBreakScarp, orc.LEFT_UPPER, 61, 64, 1., 10, [5.], [-0.6], 4096, True),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%dx%d" % (c[1], c[2], c[3]))
def test_model_matches_oracle(case):
    cls, kind, ny, nx, de, scale, params, angles, tmax, whole = case
    z = (np.cumsum(rng.standard_normal((ny, nx)), 1) * 0.05
         + rng.standard_normal((ny, nx)) * 0.02).astype(np.float32)
    xa, ya = WT.centred_axis(nx, de), WT.centred_axis(ny, de)
    A, B, C = pm.curvature_planes(z, de, de)
    for ang in angles:
        cc, sc2, ss = _plan.curvature_coefficients(ang)
        curv = cc * A - sc2 * B + ss * C
        assert np.allclose(curv, orc.directional_curvature(z, de, de, ang), rtol=1e-12, atol=1e-14)
        descs = [cls(scale, p, ang, nx, ny, de)._device_descriptor() for p in params]
        bbox = _plan.bbox_union([d["bbox"] for d in descs])
        plan = _plan.Plan(ny, nx, (0, ny, 0, nx), bbox, whole=whole, t_max=tmax)
        res = pm.match_batch_fft(curv, descs, plan, xa, ya)
        for p, d, (amp, snr) in zip(params, descs, res):
            o_amp, _, _, o_snr, det = orc.match_template(z, de, de, kind, scale, p, ang, details=True)
            _, _, n, ts = pm.synth_window(d, xa, ya)
            assert abs(n - det["n"]) < 0.5
            assert abs(ts - det["template_sum"]) <= 1e-12 * abs(ts)
            assert np.allclose(amp, o_amp, rtol=1e-8, atol=1e-10)
            assert np.allclose(snr, o_snr, rtol=1e-6, atol=1e-8)
            if ny * nx <= 70 * 70:
                a2, s2 = pm.match_one_direct(curv, d, xa, ya)
                assert np.allclose(a2, o_amp, rtol=1e-8, atol=1e-10)
                assert np.allclose(s2, o_snr, rtol=1e-6, atol=1e-8)


def test_tile_choice():
    # 10000-cell axis, 35 x 181 Scarp grid at scale 100: span 306
    T, V, nt, circ = _plan.choose_tile(10000, 306, 10000, True)
    assert (T, circ) == (2048, False) and V == 2048 - 306 and nt * V >= 10000
    # a power-of-two axis owned entirely uses its own periodicity
    assert _plan.choose_tile(2048, 306, 2048, True) == (2048, 2048, 1, True)
    # ... but not when the rank only owns part of it
    assert _plan.choose_tile(1024, 306, 2048, False)[3] is False
    with pytest.raises(ValueError):
        _plan.choose_tile(10000, 5000, 10000, True)


def test_circular_axes_have_one_origin_for_every_template():
    """A circular axis needs no halo, so its tile origin does not follow the template's support:
    T/2 for every scale - the tile (and the curvature spectra a search keeps, sc_set_option
    "spectra_mb") is then the same for all scales of a multi-scale job.  Tiled axes keep the
    support's own extent."""
    plans = [_plan.Plan(512, 512, (0, 512, 0, 512), bbox) for bbox in
             [(-12, 12, -12, 12), (-80, 79, -60, 60), (-255, 255, -200, 199)]]
    for p in plans:
        assert p.circ_y and p.circ_x and (p.Py, p.Qx) == (256, 256)
        assert p.Py >= p.bbox[1] and p.Py - p.bbox[0] <= p.Ty        # what sc_match checks
        assert p.tiles() == plans[0].tiles()
    q = _plan.Plan(900, 512, (0, 900, 0, 512), (-40, 39, -30, 30))     # tiled in y, circular in x
    assert not q.circ_y and q.circ_x and q.Py == 39 and q.Qx == 256


def test_search_grids_match_reference():
    assert len(_plan.angle_grid()) == 181 and len(_plan.age_grid()) == 35
    assert np.array_equal(_plan.angle_grid(-0.4, 0.4), orc.angle_grid(-0.4, 0.4))
    assert np.array_equal(_plan.age_grid(), orc.age_grid())


def test_single_tile_of_column_length_2048_is_replanned():
    """One tile of column length 2048 (BASELINE config C2) would send every template through the paired-template
    four-wave column pass; the planner takes three tiles of 1024 rows instead (profiles/r04_c2_plans.txt) - as long
    as that costs at most 1.6 x the cells and 1024 is the tile the axis would choose below 2048."""
    bbox = (-154, 153, -154, 153)
    p = _plan.Plan(2048, 2048, (0, 2048, 0, 2048), bbox)
    assert (p.Ty, p.nty, p.circ_y, p.Vy) == (1024, 3, False, 1024 - 307) and (p.Tx, p.ntx, p.circ_x) == (2048, 1, True)
    assert p.nty * p.Vy >= 2048 and p.Py == bbox[1] and p.Qx == 1024
    # small supports (512 would be the axis' choice) and supports 1024 cannot hold keep the circular tile
    for bb in ((-30, 30, -30, 30), (-600, 600, -154, 153)):
        q = _plan.Plan(2048, 2048, (0, 2048, 0, 2048), bb)
        assert (q.Ty, q.nty, q.circ_y) == (2048, 1, True), q
    # several tiles already: nothing to re-plan
    r = _plan.Plan(10000, 10000, (0, 10000, 0, 10000), bbox)
    assert (r.Ty, r.Tx, r.nty, r.ntx) == (2048, 2048, 6, 6)
    # the cap of an explicit t_max is respected
    assert _plan.Plan(2048, 2048, (0, 2048, 0, 2048), bbox, t_max=1024).Ty == 1024


# ---- the wave-per-column pass's stage 2 -> stage 3 exchange across lanes (sc_fft.hip: xlane_transpose4, SC_I1_XLANE) ----------
def _xlane_model(T):
    """Lanes as numpy arrays: the Stockham placement of stage 2 (radix 16, stride 16) of one wave's line, the 4 x 4
    transposes between the 16-lane row and m & 3 that v_permlane32_swap / v_permlane16_swap perform, and the register
    the kernel then hands to input j of stage-3 butterfly b of set u3.  Returns (what the kernel feeds, what the LDS
    path reads) as element indices of the line."""
    S, U = T // 16, T // 1024                    # 16-point sets per line, sets per lane
    R3, NB3 = T // 256, 16 // (T // 256)         # stage 3: radix, butterflies per set
    lanes = np.arange(64)
    # stage 2: output m of lane L's set u lies at element (bt & 15) + 16 m + 256 (bt >> 4), bt = L + 64 u
    o2 = np.empty((U, 16, 64), dtype=int)
    for u in range(U):
        bt = lanes + 64 * u
        for m in range(16):
            o2[u, m] = (bt & 15) + 16 * m + 256 * (bt >> 4)

    def swap32(a, b):                            # a's lanes 32-63 <-> b's lanes 0-31
        a, b = a.copy(), b.copy()
        a[32:], b[:32] = b[:32].copy(), a[32:].copy()
        return a, b

    def swap16(a, b):                            # a's odd rows of 16 lanes <-> b's even rows
        a, b = a.copy(), b.copy()
        for r in (1, 3):
            lo, hi = 16 * r, 16 * r + 16
            elo, ehi = 16 * (r - 1), 16 * (r - 1) + 16
            a[lo:hi], b[elo:ehi] = b[elo:ehi].copy(), a[lo:hi].copy()
        return a, b

    for u in range(U):
        for mh in range(4):
            r = [o2[u, 4 * mh + k] for k in range(4)]
            r[0], r[2] = swap32(r[0], r[2])
            r[1], r[3] = swap32(r[1], r[3])
            r[0], r[1] = swap16(r[0], r[1])
            r[2], r[3] = swap16(r[2], r[3])
            for k in range(4):
                o2[u, 4 * mh + k] = r[k]
    fed = np.empty((U, 16, 64), dtype=int)
    want = np.empty((U, 16, 64), dtype=int)
    for u3 in range(U):
        for b in range(NB3):
            for j in range(R3):
                fed[u3, b + NB3 * j] = o2[j >> 2, 4 * (u3 + U * b) + (j & 3)]
                # set_load of stage 3: a[i] = element tt + i S, tt = lane + 64 u3; butterfly b takes a[b + NB3 j]
                want[u3, b + NB3 * j] = (lanes + 64 * u3) + (b + NB3 * j) * S
    return fed, want


def test_cross_lane_exchange_feeds_stage_3_what_the_lds_path_reads():
    """k_inv_cols_w8 (round 4) hands stage 2's outputs to stage 3 through v_permlane swaps instead of the LDS
    (profiles/r04_xlane_exchange.txt).  The GPU suite checks the result bit for bit against the four-column kernels;
    this is the index algebra on its own - for column lengths 2048 (16 x 16 x 8) and 1024 (16 x 16 x 4) every
    stage-3 input register must hold exactly the element the LDS path would have read."""
    for T in (2048, 1024):
        fed, want = _xlane_model(T)
        assert np.array_equal(fed, want), T
        assert len(np.unique(fed)) == fed.size == T          # every element of the line exactly once
