"""The multi-GPU decomposition on ONE GPU: every rank's halo-extended block is
searched on its own (non-periodic block, window limits in global coordinates)
and the stitched tiles must reproduce the whole-DEM search.  Also runs the RCCL
entry points with a single-rank communicator (periodic self-images only)."""
import os
import sys

import numpy as np
import pytest

import scarplet_oracle as orc
import scarplet_amd as sl
from scarplet_amd import _lib, _plan, dist as sd, synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P_AMP = orc.PARITY["amp"]


def whole_and_tiled(g, Template, scale, params, angles, nranks, method, backend="host"):
    z = g._griddata
    ny, nx = z.shape
    m = sl.Matcher(g)
    whole = m.search(Template, scale, params, angles, method=method).result()
    arr, bbox, area = m.describe(Template, scale, np.asarray(params, float), np.asarray(angles, float))
    halo = sd.halo_for_search(bbox, ny, nx)
    py, px = sd.grid_dims(nranks, ny, nx)
    lay = sd.Layout(ny, nx, py, px, halo)
    out = [np.zeros((ny, nx)) for _ in range(4)]
    for r in range(nranks):
        c = lay.core(r)
        mt = sl.Matcher()
        mt.ny, mt.nx, mt.de = ny, nx, g._georef_info.dx
        blk = sd.assemble_block_reference(z, lay, r)
        mt.set_block(np.ascontiguousarray(blk), lay.block_origin(r), (ny, nx), c,
                     g._georef_info.dx, g._georef_info.dy)
        mt.plan, sp = mt.plan_for(bbox, area, method, None, n_params=len(params))
        mt.ctx.reset_best()
        mt.ctx.match(arr, sp)
        mt.params, mt.angles = np.asarray(params, float), np.asarray(angles, float)
        res = mt.result()
        for k in range(4):
            out[k][c[0]:c[1], c[2]:c[3]] = res[k]
    return whole, out


@pytest.mark.parametrize("method", ["fft", "direct"])
@pytest.mark.parametrize("nranks,shape", [(2, (150, 131)), (4, (200, 260)), (8, (256, 300))])
def test_tiled_blocks_reproduce_whole_dem(method, nranks, shape):
    g = synthetic.synthetic_scarp(shape[1], seed=nranks, ny=shape[0])
    params = [2.0, 10.0, 50.0]
    angles = _plan.angle_grid(-1.2, 1.2)[::12]
    whole, tiled = whole_and_tiled(g, sl.Scarp, 12, params, angles, nranks, method)
    same = (whole[1] == tiled[1]) & (whole[2] == tiled[2])
    assert same.mean() > 0.995, float(same.mean())
    assert np.allclose(whole[0][same], tiled[0][same], rtol=2e-4, atol=2e-6 * np.abs(whole[0]).max())
    assert np.allclose(whole[3][same], tiled[3][same], rtol=2e-3, atol=2e-6 * whole[3].max())
    # the few differing cells are near-ties: same SNR within tolerance
    assert np.allclose(whole[3][~same], tiled[3][~same], rtol=orc.PARITY["tie_rtol"])


def test_tiled_blocks_against_oracle():
    g = synthetic.synthetic_scarp(120, seed=5, ny=96)
    z = g._griddata
    params, angles = [3.0, 20.0], _plan.angle_grid(-0.5, 0.5)[::4]
    _, tiled = whole_and_tiled(g, sl.Scarp, 10, params, angles, 4, "fft")
    a_st, s_st = orc.snr_stack(z, 1.0, 1.0, orc.SCARP, 10, params, angles)
    T = len(params) * len(angles)
    chk = orc.check_fold(tiled, a_st.reshape(T, 96, 120), s_st.reshape(T, 96, 120),
                         np.repeat(params, len(angles)), np.tile(angles, len(params)),
                         tie_rtol=orc.PARITY["tie_rtol"], amp_tol=(2e-4, 2e-6 * np.abs(a_st).max()),
                         snr_tol=(2e-3, 2e-6 * s_st.max()))
    print("tiled blocks vs oracle: inexact=%d below=%d exact=%.6f tie=%d of %d"
          % (chk["n_inexact"], chk["n_below_only"], chk["exact_frac"], chk["n_tie"], chk["n"]))
    assert chk["n_bad"] == 0, chk
    assert chk["exact_frac"] >= 0.99, chk


def test_rccl_single_rank_halo_exchange():
    """sc_comm_init / sc_halo_exchange / sc_set_dem_device with one rank: the
    halo consists of periodic images of the rank's own core."""
    g = synthetic.synthetic_scarp(140, seed=9, ny=100)
    z = g._griddata
    params, angles = [5.0, 30.0], _plan.angle_grid(-0.3, 0.3)[::6]
    m = sl.Matcher(g)
    whole = m.search(sl.Scarp, 10, params, angles, method="fft").result()
    dm = sd.DistMatcher(0, 1, z.shape, 1.0, 1.0, device=0, backend="rccl",
                        broadcast_bytes=lambda b: b)
    dm.m.ctx.comm_init(dm.m.ctx.comm_unique_id(), 0, 1)
    res = dm.search(sl.Scarp, 10, params, angles, z, method="fft").result()
    for k in range(4):
        assert np.allclose(res[k], whole[k], rtol=2e-3, atol=1e-6 * np.abs(whole[k]).max() + 1e-12)
    # final gather (sc_gather_result): with one rank the root places its own planes
    # (exact mode, the default: the record that travels carries the float64 argmax with amp / snr ROUNDED to float32 -
    #  every rank's cells alike - where result() lays the float64 values over its own planes)
    full = dm.gather(0)
    assert np.array_equal(full[1], res[1]) and np.array_equal(full[2], res[2])
    for k in (0, 3):
        assert np.allclose(full[k], res[k], rtol=1e-7, atol=0)
    dm.search(sl.Scarp, 10, params, angles, z, method="fft", exact=False)
    full, res = dm.gather(0), dm.result()
    for k in range(4):
        assert np.array_equal(full[k], res[k])


def test_gather_result_places_a_core_inside_the_dem():
    """sc_gather_result without a communicator: a context that owns a sub-rectangle
    of the DEM gets its four planes at that place of the full arrays, zeros elsewhere."""
    g = synthetic.synthetic_scarp(150, seed=4, ny=120)
    z = g._griddata
    params, angles = [4.0], _plan.angle_grid(-0.2, 0.2)[::5]
    lay = sd.Layout(120, 150, 1, 2, sd.halo_for_search((-20, 20, -20, 20), 120, 150))
    core = lay.core(1)
    m = sl.Matcher()
    m.set_block(sd.assemble_block_reference(z, lay, 1), lay.block_origin(1), z.shape, core, 1.0, 1.0)
    res = m.search(sl.Scarp, 8, params, angles, method="fft").result()
    # rank numbering: pretend to be rank 0 of 1 with that core
    out = m.ctx.gather_result(0, [core], z.shape, np.repeat(params, len(angles)),
                              np.tile(angles, len(params)), True)
    for k in range(4):
        assert np.array_equal(out[k][core[0]:core[1], core[2]:core[3]], res[k])
        mask = np.ones(z.shape, bool)
        mask[core[0]:core[1], core[2]:core[3]] = False
        assert not out[k][mask].any()


@pytest.mark.parametrize("partition", ["grid", "tiles"])
def test_c4_eight_rank_blocks_at_full_size(partition):
    """BASELINE config C4 on one GPU: the 10000 x 10000 DEM cut into eight rank cores - the even
    2 x 4 grid (5000 x 2500 cores), or dist.tile_cores' rectangles of whole FFT tiles (four
    1 x 5 strips around four 2 x 2 blocks of 1741-cell tiles) - every rank's block (core + torus
    halo, as sc_halo_exchange assembles it) searched on its own with all 35 ages, the cores
    stitched and compared with the whole-DEM search.  The runs tile differently (grid blocks:
    3 x 4 tiles of 2048 x 1024, whole DEM: 6 x 6 of 2048^2), so they differ by float32
    rounding: winners must agree except inside the tie window."""
    n = 10000
    g = synthetic.synthetic_scarp(n)
    z = g._griddata
    ages = _plan.age_grid()
    angles = _plan.angle_grid()[[3, 30, 61, 88, 117, 150, 176]]
    m = sl.Matcher(g)
    whole = m.search(sl.Scarp, 100, ages, angles, method="fft").result_array()
    arr, bbox, area = m.describe(sl.Scarp, 100, ages, angles)
    halo = sd.halo_for_search(bbox, n, n)
    py, px = sd.grid_dims(8, n, n)
    assert (py, px) in ((2, 4), (4, 2))
    cores = sd.tile_cores(8, n, n, bbox) if partition == "tiles" else None
    assert (cores is not None) == (partition == "tiles")
    lay = sd.Layout(n, n, py, px, halo, cores=cores)
    tiled = np.zeros_like(whole)
    for r in range(8):
        c = lay.core(r)
        blk = np.ascontiguousarray(sd.assemble_block_reference(z, lay, r))
        m.set_block(blk, lay.block_origin(r), (n, n), c, 1.0, 1.0)
        plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
        m.ctx.reset_best()
        m.ctx.match(arr, sp)
        tiled[:, c[0]:c[1], c[2]:c[3]] = m.ctx.get_result(np.repeat(ages, len(angles)), np.tile(angles, len(ages)))
    same = (whole[1] == tiled[1]) & (whole[2] == tiled[2])
    tie = orc.PARITY["tie_rtol"]
    print("C4 blocks (" + partition + ") vs whole DEM: same (age, angle) %.6f of %d cells; the rest within the tie window: %s"
          % (same.mean(), same.size, bool(np.allclose(whole[3][~same], tiled[3][~same], rtol=tie))))
    assert same.mean() > 0.999, float(same.mean())
    assert np.allclose(whole[3][~same], tiled[3][~same], rtol=tie)
    assert np.allclose(whole[0][same], tiled[0][same], rtol=2e-4, atol=2e-6 * np.abs(whole[0]).max())
    assert np.allclose(whole[3][same], tiled[3][same], rtol=2e-3, atol=2e-6 * whole[3].max())
    if partition != "tiles":
        return
    # The EXACT mode (round 6; what the 8-GPU bench line times): every block settles its own near-ties in float64
    # (sc_settle_exact on the halo-extended block).  Blocks and whole DEM tile differently - other float32 errors, other
    # near-tie lists - but the float64 argmax is one: the same (age, orientation) in EVERY cell, and the settled cells'
    # float64 values agree to the last bits (the planes they are scored on are the same numbers).
    m.set_data(g)
    whole_x = np.array(m.search(sl.Scarp, 100, ages, angles, method="fft", exact=True).result_array())
    st_whole = dict(m.exact_stats)
    tiled_x = np.zeros_like(whole_x)
    flagged = 0
    for r in range(8):
        c = lay.core(r)
        blk = np.ascontiguousarray(sd.assemble_block_reference(z, lay, r))
        m.set_block(blk, lay.block_origin(r), (n, n), c, 1.0, 1.0)
        plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
        st = m.run_described(arr, sp, m.exact_window_for(arr, sp), m.end_twins(arr, len(ages), angles))
        flagged += st["flagged_cells"]
        tiled_x[:, c[0]:c[1], c[2]:c[3]] = m.ctx.get_result(np.repeat(ages, len(angles)), np.tile(angles, len(ages)))
    same_x = (whole_x[1] == tiled_x[1]) & (whole_x[2] == tiled_x[2])
    print("C4 blocks, exact: same (age, angle) in %d of %d cells (float32 mode: %d differ); flagged %d in blocks, %d whole"
          % (int(same_x.sum()), same_x.size, int((~same).sum()), flagged, st_whole["flagged_cells"]))
    assert same_x.all(), int((~same_x).sum())


# ---- orientation sharding (dist.OrientationMatcher) ------------------------------------------
def _fold_raw(parts):
    """sc_fold_ranks on the host: per cell the greatest SNR, equal SNRs to the smaller id; the
    amplitude follows the winner.  parts: [(amp, snr, id)] float32 / float32 / uint32."""
    amp, snr, idx = [np.array(a, copy=True) for a in parts[0]]
    for (a, s, i) in parts[1:]:
        ks = (snr.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idx.astype(np.uint64))
        kt = (s.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - i.astype(np.uint64))
        take = kt > ks
        amp[take], snr[take], idx[take] = a[take], s[take], i[take]
    return amp, snr, idx


@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("method", ["fft", "direct"])
def test_orientation_chunks_fold_to_the_single_search_bit_for_bit(nranks, method):
    """Every rank's chunk of the orientation grid searched on its own (whole DEM, ids in fold
    order, the plan of the whole grid) and the records folded as sc_fold_ranks folds them:
    amplitude, SNR and winner id equal the single search's in every bit - exact ties
    included (a noise-free surface has many)."""
    from scarplet_amd import WindowedTemplate as WT
    for (g, Template, scale, params) in [
            (synthetic.synthetic_scarp(230, seed=4, ny=200), sl.Scarp, 14, [1.0, 4.0, 20.0, 100.0]),
            (synthetic.synthetic_scarp(128, seed=2, sigma=0.0), sl.Scarp, 10, [1.0, 4.0, 20.0, 100.0]),
            (synthetic.synthetic_scarp(150, seed=6, ny=140), WT.Channel, 8, [0.05, 0.1, 0.2]),
            (synthetic.synthetic_scarp(150, seed=7, ny=160), WT.LeftFacingUpperBreakScarp, 12, [2.0, 30.0])]:
        angles = _plan.angle_grid(-np.pi / 2, np.pi / 2)[::17]          # 11 orientations
        om = sd.OrientationMatcher(0, 1, g)
        ctx, n_par = om.m.ctx, len(params)
        mine, sp = om.describe(Template, scale, params, angles, method=method)
        assert len(mine) == n_par * len(angles)
        om.run(mine, sp)
        whole = ctx.get_best()
        whole_arr = om.result_array()
        parts, arrs = [], []
        for (b0, b1) in sd.orientation_chunks(len(angles), nranks):
            ctx.reset_best()
            if b1 > b0:
                sub = (type(mine[0]) * ((b1 - b0) * n_par)).from_buffer(om._keep, b0 * n_par * _lib.C.sizeof(type(mine[0])))
                ctx.match(sub, sp)
            parts.append(ctx.get_best())
            arrs.append(om.m.result_array())
        amp, snr, idx = _fold_raw(parts)
        assert np.array_equal(idx, whole[2])
        assert np.array_equal(snr.view(np.uint32), whole[1].view(np.uint32))
        assert np.array_equal(amp.view(np.uint32), whole[0].view(np.uint32))
        # and through the host fold of the decoded (amp, age, angle, snr) planes
        assert np.array_equal(sd.fold_host(arrs), whole_arr)
        # the ids decode in fold order: the single search through Matcher gives the same maps
        ref = sl.Matcher(g).search(Template, scale, params, angles, method=method).result_array()
        assert np.array_equal(ref, whole_arr)


def test_fold_ranks_with_a_one_rank_communicator():
    """sc_fold_ranks end to end on the hardware there is: pack, ncclAllReduce(max, uint64),
    unpack, ncclAllReduce(sum, float) - with one rank the record must come back unchanged,
    cells nobody won (id 0xFFFFFFFF, window-limit border) included."""
    g = synthetic.synthetic_scarp(150, seed=8, ny=140)
    om = sd.OrientationMatcher(0, 1, g)
    ctx = om.m.ctx
    ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    mine, sp = om.describe(sl.Scarp, 12, [2.0, 30.0], _plan.angle_grid(-1.0, 1.0)[::20], method="fft")
    ctx.reset_best()
    ctx.match(mine, sp)
    before = ctx.get_best()
    assert (before[2] == 0xFFFFFFFF).any() and (before[2] != 0xFFFFFFFF).any()
    ctx.fold_ranks()
    after = ctx.get_best()
    for a, b in zip(before, after):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _fold_twin(ang):
    """the orientation plane with the grid's end twins as one (+pi/2 is -pi/2 for the symmetric built-ins)"""
    a = np.array(ang, dtype=float)
    a[np.abs(a - np.pi / 2) < 1e-12] = -np.pi / 2
    return a


@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("method", ["fft", "direct"])
def test_exact_orientation_sharding_equals_the_single_context(nranks, method):
    """Exact mode of the orientation sharding with the REAL device code, the ranks as threads with a context each on this
    GPU (tools/thread_transport.py, host backend): match with the window on, sc_snapshot_best, the fold (sc_get_best ->
    fold_records -> sc_set_best), sc_rank_candidates, the exchange, sc_settle_pairs.  Every rank ends with the same record
    in every bit, and it names the same (age, orientation) in every cell as ONE context searching all the templates with
    exact=True (sc_settle_exact) - the near-ties between templates of different ranks, which no rank's event list holds,
    included (a noise-free surface is full of them)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from thread_transport import run_ranks
    grid = _plan.angle_grid(-np.pi / 2, np.pi / 2)                             # both ends: the twins sit on the first and last rank
    # every orientation twice, 2e-5 rad apart - near-ties inside the float32 error - and listed so that the contiguous chunks
    # put the two of a pair on DIFFERENT ranks (2, 3 and 8 of them); the grid's own end twins first and last
    base = grid[:-1:23]
    paired = np.concatenate([base, base[1:] + 2e-5, grid[-1:]])
    cases = [(synthetic.synthetic_scarp(230, seed=4, ny=200), sl.Scarp, orc.SCARP, 14, [1.0, 4.0, 4.0004, 20.0, 100.0], paired),
             (synthetic.synthetic_scarp(160, seed=2, sigma=0.0), sl.Scarp, orc.SCARP, 10, [1.0, 4.0, 4.05, 20.0], grid[::3]),
             (synthetic.synthetic_scarp(150, seed=6, ny=140), sl.Ricker, orc.RICKER, 8, [2.0, 4.0, 4.0003], paired)]
    teeth = 0
    for (g, Template, kind, scale, params, angles) in cases:
        single = sl.Matcher(g, ctx=_lib.Context(0))
        want = np.stack(single.search(Template, scale, params, angles, method=method, exact=True).result())
        want_rec = [a.copy() for a in single.ctx.get_best()]
        st1 = dict(single.exact_stats)
        assert st1.get("route") == "device", st1
        f32 = np.stack(single.search(Template, scale, params, angles, method=method, exact=False).result())

        def rank_body(rank, transport):
            m = sl.Matcher(g, ctx=_lib.Context(0))
            om = sd.OrientationMatcher(rank, nranks, None, backend="host", transport=transport, matcher=m)
            om.search(Template, scale, params, angles, method=method)           # (exact by default)
            return [a.copy() for a in m.ctx.get_best()], np.array(om.result_array()), dict(om.exact_stats), method

        outs = run_ranks(nranks, rank_body)
        rec0, arr0, st, used = outs[0]
        print("%s %s, %d ranks (%s): one context %s; sharded %s" % (Template.__name__, g._griddata.shape, nranks, used, st1, st))
        for (rec, arr, st_r, _) in outs[1:]:
            assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(rec, rec0))
            assert np.array_equal(arr, arr0, equal_nan=True) and st_r == st
        # the same (age, orientation) in every cell as the single context's settle (the grid's end twins are one maximum) ...
        off = (arr0[1] != want[1]) | (_fold_twin(arr0[2]) != _fold_twin(want[2]))
        assert not off.any(), int(off.sum())
        # ... (the test has teeth: the float32 fold alone names other templates in some cells)
        n_f32 = int(((f32[1] != want[1]) | (_fold_twin(f32[2]) != _fold_twin(want[2]))).sum())
        assert n_f32 == st1["changed_cells"], (n_f32, st1)
        teeth += n_f32
        # ... and (amp, snr) within the float32 paths' error of it (a cell settled here and not there carries float64 against
        # float32 of the same template; twins differ in the amplitude's sign)
        rtol, afac = orc.snr_tolerance(kind)
        assert np.allclose(arr0[3], want[3], rtol=rtol, atol=afac * np.nanmax(want[3]), equal_nan=True)
        same = arr0[2] == want[2]
        a_tol = P_AMP[0] * np.abs(want[0]) + P_AMP[1] * np.nanmax(np.abs(want[0]))
        assert (np.abs(arr0[0] - want[0])[same] <= a_tol[same]).all()
        assert (np.abs(np.abs(arr0[0]) - np.abs(want[0])) <= a_tol).all()
    assert teeth >= 20, teeth


@pytest.mark.parametrize("with_comm", [False, True])
def test_exact_orientation_exchange_on_the_device(with_comm):
    """The RCCL form of the exchange on the hardware there is: the candidate list stays on the device, sc_exchange_candidates
    all-gathers the counts and the padded slots (a one-rank communicator runs both ncclAllGather; without one the union is
    the rank's own list) and sc_settle_pairs settles the device list - the same record, bit for bit, as the host form of
    the exchange (ctx.rank_candidates() handed back to sc_settle_pairs) and the same (age, orientation) as
    sc_settle_exact."""
    g = synthetic.synthetic_scarp(230, seed=4, ny=200)
    grid = _plan.angle_grid(-np.pi / 2, np.pi / 2)
    base = grid[:-1:23]
    angles = np.concatenate([base, base[1:] + 2e-5, grid[-1:]])
    params = [1.0, 4.0, 4.0004, 20.0]
    ctx = _lib.Context(0)
    om = sd.OrientationMatcher(0, 1, None, matcher=sl.Matcher(g, ctx=ctx))          # backend "rccl"
    if with_comm:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
    om.search(sl.Scarp, 14, params, angles, method="fft")
    st = dict(om.exact_stats)
    rec = [a.copy() for a in ctx.get_best()]
    arr = np.array(om.result_array())
    assert st["changed_cells"] > 20 and st["route"].startswith("device"), st
    # the host form of the exchange on the same context
    mine, sp = om.describe(sl.Scarp, 14, params, angles, method="fft")
    ctx.reset_best()
    ctx.set_option("near_window", om.m.exact_window_for(om._keep, sp))
    ctx.match(mine, sp)
    ctx.set_option("near_window", 0.0)
    ctx.snapshot_best()
    ctx.fold_ranks()
    pairs = ctx.rank_candidates()
    assert len(pairs) == ctx.rank_candidates(fetch=False) > 0
    st2 = ctx.settle_pairs(om._keep, pairs, om.m.end_twins(om._keep, len(params), angles), om.m.EXACT_MAX_F64)
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(rec, ctx.get_best()))
    assert {k: v for k, v in st.items() if k != "route"} == st2
    # and one context's own settle
    want = np.stack(sl.Matcher(g, ctx=ctx).search(sl.Scarp, 14, params, angles, method="fft", exact=True).result())
    assert np.array_equal(arr[1], want[1]) and np.array_equal(_fold_twin(arr[2]), _fold_twin(want[2]))
    if with_comm:
        ctx.comm_destroy()
