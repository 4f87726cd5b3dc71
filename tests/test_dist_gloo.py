"""Multi-rank host logic on CPU: tile layout, torus halo transfer lists and
their execution over torch.distributed/gloo with world_size 2 and 4."""
import multiprocessing as mp
import os
import socket

import sys

import numpy as np
import pytest

from scarplet_amd import dist as sd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_tiles_the_dem():
    for (n, ny, nx) in [(2, 100, 90), (4, 64, 200), (8, 10000, 10000), (6, 51, 47)]:
        py, px = sd.grid_dims(n, ny, nx)
        assert py * px == n
        lay = sd.Layout(ny, nx, py, px, (3, 4, 5, 6))
        cover = np.zeros((ny, nx), int)
        for r in range(n):
            c = lay.core(r)
            cover[c[0]:c[1], c[2]:c[3]] += 1
        assert (cover == 1).all()
    assert sd.grid_dims(8, 10000, 10000) in ((2, 4), (4, 2))


def test_transfer_list_fills_every_block():
    """Executing the global list on host arrays reproduces DEM-modulo-size
    blocks for every rank (includes periodic self-images when a rank owns a
    whole axis)."""
    rng = np.random.default_rng(5)
    for (n, ny, nx, halo) in [(2, 40, 37, (7, 6, 9, 8)), (4, 30, 44, (5, 5, 12, 3)),
                              (1, 20, 21, (4, 3, 2, 5)), (8, 64, 96, (10, 9, 11, 12))]:
        z = rng.standard_normal((ny, nx))
        py, px = sd.grid_dims(n, ny, nx)
        lay = sd.Layout(ny, nx, py, px, halo)
        blocks = []
        for r in range(n):
            b = np.full(lay.block_shape(r), np.nan)
            c = lay.core(r)
            b[halo[0]:halo[0] + c[1] - c[0], halo[2]:halo[2] + c[3] - c[2]] = z[c[0]:c[1], c[2]:c[3]]
            blocks.append(b)
        for (src, dst, sy0, sx0, dy0, dx0, h, w) in lay.transfers():
            piece = blocks[src][sy0:sy0 + h, sx0:sx0 + w]
            assert not np.isnan(piece).any(), "source rectangle outside the source core"
            blocks[dst][dy0:dy0 + h, dx0:dx0 + w] = piece
        for r in range(n):
            assert np.array_equal(blocks[r], sd.assemble_block_reference(z, lay, r))
        # per-rank views pair up: every send has its receive, in the same order
        for a in range(n):
            for b in range(n):
                if a == b:
                    continue
                sends = [x[-1] for x in lay.rank_transfers(a, True) if x[1] == 1 and x[0] == b]
                recvs = [x[-1] for x in lay.rank_transfers(b, True) if x[1] == 0 and x[0] == a]
                assert sends == recvs


def test_tile_partition_balances_tile_counts():
    """36 tiles over 8 ranks: five per rank at most (four 1x5 strips around four 2x2 blocks),
    where the even 2x4 grid gives every rank six."""
    for (n, nty, ntx, want) in [(8, 6, 6, 5), (2, 6, 6, 18), (4, 6, 6, 9), (6, 6, 6, 6), (3, 2, 2, 2),
                                (5, 1, 7, 2), (4, 2, 2, 1)]:
        part = sd.tile_partition(n, nty, ntx)
        assert len(part) == n
        cover = np.zeros((nty, ntx), int)
        for (a0, a1, b0, b1) in part:
            cover[a0:a1, b0:b1] += 1
        assert (cover == 1).all()
        assert max((a1 - a0) * (b1 - b0) for (a0, a1, b0, b1) in part) == want
    assert sd.tile_partition(5, 2, 2) is None
    bbox = (-154, 153, -154, 153)
    cores = sd.tile_cores(8, 10000, 10000, bbox)
    assert cores is not None and len(cores) == 8
    lay = sd.Layout(10000, 10000, 2, 4, sd.halo_for_search(bbox, 10000, 10000), cores=cores)
    from scarplet_amd import _plan
    padded = []                       # padded cells of each rank's own plan, in 2048^2 tiles
    for r in range(8):
        c = lay.core(r)
        ty, _, nty, _ = _plan.choose_tile(c[1] - c[0], 307, 10000, False)
        tx, _, ntx, _ = _plan.choose_tile(c[3] - c[2], 307, 10000, False)
        padded.append(ty * nty * tx * ntx / 2048.0 ** 2)
    assert max(padded) <= 5 and sum(padded) <= 36
    # no gain over the even grid: keep the grid
    assert sd.tile_cores(4, 10000, 10000, bbox) is None
    assert sd.tile_cores(2, 10000, 10000, bbox) is None
    with pytest.raises(ValueError):
        sd.Layout(10, 10, 1, 2, (1, 1, 1, 1), cores=[(0, 10, 0, 6), (0, 10, 5, 10)])
    with pytest.raises(ValueError):
        sd.Layout(10, 10, 1, 2, (1, 1, 1, 1), cores=[(0, 10, 0, 5), (0, 9, 5, 10)])


def _uneven_cores(ny, nx):
    """A pinwheel of five rectangles (not a grid: no cut runs through the whole DEM)."""
    a, b = ny // 3, nx // 3
    return [(0, a, 0, nx - b), (0, ny - a, nx - b, nx), (ny - a, ny, b, nx), (a, ny, 0, b),
            (a, ny - a, b, nx - b)]


def test_transfer_list_fills_uneven_blocks():
    rng = np.random.default_rng(6)
    for (ny, nx, halo) in [(40, 37, (7, 6, 9, 8)), (33, 60, (10, 10, 12, 3))]:
        z = rng.standard_normal((ny, nx))
        cores = _uneven_cores(ny, nx)
        lay = sd.Layout(ny, nx, 1, len(cores), halo, cores=cores)
        n = lay.nranks
        blocks = []
        for r in range(n):
            b = np.full(lay.block_shape(r), np.nan)
            c = lay.core(r)
            b[halo[0]:halo[0] + c[1] - c[0], halo[2]:halo[2] + c[3] - c[2]] = z[c[0]:c[1], c[2]:c[3]]
            blocks.append(b)
        for (src, dst, sy0, sx0, dy0, dx0, h, w) in lay.transfers():
            piece = blocks[src][sy0:sy0 + h, sx0:sx0 + w]
            assert not np.isnan(piece).any()
            blocks[dst][dy0:dy0 + h, dx0:dx0 + w] = piece
        for r in range(n):
            assert np.array_equal(blocks[r], sd.assemble_block_reference(z, lay, r))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ny, nx, halo, q):
    try:
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from torch_transport import TorchTransport
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port,
                                rank=rank, world_size=world)
        z = np.random.default_rng(9).standard_normal((ny, nx))
        py, px = sd.grid_dims(world, ny, nx)
        lay = sd.Layout(ny, nx, py, px, halo, cores=_uneven_cores(ny, nx) if world == 5 else None)
        c = lay.core(rank)
        blk = sd.exchange_host(z[c[0]:c[1], c[2]:c[3]], lay, rank, TorchTransport())
        ok = np.array_equal(blk, sd.assemble_block_reference(z, lay, rank))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
    except Exception as e:                       # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world,ny,nx,halo", [(2, 48, 50, (9, 8, 7, 10)), (4, 40, 60, (6, 6, 13, 5)),
                                              (5, 45, 39, (8, 7, 9, 6))])      # 5: uneven pinwheel cores
def test_halo_exchange_over_gloo(world, ny, nx, halo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ny, nx, halo, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)], res


class _OracleContext(object):
    """Stands in for the GPU context in the CPU test of OrientationMatcher: `match` folds the
    oracle's maps of the templates it is handed (ties keep the incumbent, like sc_match)."""

    def __init__(self, m, z, kind, scale):
        self.m, self.z, self.kind, self.scale = m, z, kind, scale

    def reset_best(self):
        ny, nx = self.z.shape
        self.rec = np.zeros((4, ny, nx))

    def match(self, templates, plan, sync=True):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import scarplet_oracle as orc
        for t in templates:
            age, ang = float(self.m._id_par[t.id]), float(self.m._id_ang[t.id])
            amp, _, _, snr = orc.match_template(self.z, 1.0, 1.0, self.kind, self.scale, age, ang)
            take = snr > self.rec[3]
            for k, v in enumerate((amp, age, ang, snr)):
                self.rec[k] = np.where(take, v, self.rec[k])


def _orientation_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from torch_transport import TorchTransport
        import scarplet_oracle as orc
        import scarplet_amd as sl
        from scarplet_amd.core import Matcher
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        rng = np.random.default_rng(17)                      # the same DEM on every rank
        z = (np.cumsum(rng.standard_normal((40, 36)), 1) * 0.05 + rng.standard_normal((40, 36)) * 0.03)
        z[:, 18:] = z[:, :18][:, ::-1]                       # mirror symmetry: exact SNR ties between orientations
        ages, angles = [1.0, 4.0, 16.0], np.linspace(-1.2, 1.2, 5)
        m = object.__new__(Matcher)                          # host side only: descriptors and plans need no device
        m.ny, m.nx, m.de, m.core, m.whole = 40, 36, 1.0, (0, 40, 0, 36), True
        m.ctx = _OracleContext(m, z, orc.SCARP, 6)
        m.result_array = lambda: m.ctx.rec
        om = sd.OrientationMatcher(rank, world, None, backend="host", transport=TorchTransport(), matcher=m)
        om.search(sl.Scarp, 6, ages, angles, method="fft", exact=False)
        ok = True
        if rank == 0:
            # one context folding every template in id order (orientation-major), ties to the incumbent
            want = np.zeros((4, 40, 36))
            for ang in angles:
                for age in ages:
                    amp, _, _, snr = orc.match_template(z, 1.0, 1.0, orc.SCARP, 6, age, float(ang))
                    take = snr > want[3]
                    for k, v in enumerate((amp, age, float(ang), snr)):
                        want[k] = np.where(take, v, want[k])
            ok = bool(np.array_equal(om.result_array(), want))
            # every rank got its contiguous chunk of the orientation grid, all ages each
            ch = sd.orientation_chunks(len(angles), world)
            ok = ok and sum(b - a for a, b in ch) == len(angles)
        else:
            ok = om.result_array() is None
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok))
    except Exception as e:                       # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_orientation_sharded_search_folds_over_gloo(world):
    """OrientationMatcher end to end on CPU ranks (host backend): every rank describes the whole
    grid, searches its chunk of the orientations (the oracle answers for the GPU), the records are
    gathered through the transport and folded in rank order - bit for bit the record of ONE context
    folding all templates in id order, exact SNR ties included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_orientation_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)], res


# ---- exact mode of the orientation sharding: the candidates' exchange and the settle of the union ----------------------
NONE_ID = 0xFFFFFFFF


def _jitter(tid, shape):
    """A deterministic relative error in [-1, 1] per (template, cell): what stands in for the float32 paths' rounding."""
    ny, nx = shape
    c = np.arange(ny * nx, dtype=np.uint64).reshape(ny, nx)
    h = (c * np.uint64(2654435761) + np.uint64(int(tid) * 40503 + 977)) % np.uint64(10007)
    return h.astype(np.float64) / 5003.0 - 1.0


class _ExactOracleContext(object):
    """The device side of OrientationMatcher's exact mode on the CPU: float32 scores = the oracle's float64 SNR with a
    relative error up to `noise` (well above float32 rounding, below half the near-tie window), a float32 record with
    near-tie events as sc_match lists them (cell, template scored, holder, the larger score), and the calls the ranks make
    around the fold: snapshot_best, get_best / set_best, rank_candidates, settle_pairs (float64 = the oracle itself)."""

    def __init__(self, m, z, kind, scale, noise):
        self.m, self.z, self.kind, self.scale, self.noise = m, z, kind, scale, noise
        self.opt, self.cache, self.patch = {}, {}, {}

    def core_shape(self):
        return self.z.shape

    def set_option(self, key, value):
        self.opt[key] = value

    def reset_best(self):
        self.amp = np.zeros(self.z.shape, np.float32)
        self.snr = np.zeros(self.z.shape, np.float32)
        self.idx = np.full(self.z.shape, NONE_ID, np.uint32)
        self.events, self.patch, self.w_used = [], {}, 0.0

    def _f64(self, tid):
        if tid not in self.cache:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import scarplet_oracle as orc
            amp, _, _, snr = orc.match_template(self.z, 1.0, 1.0, self.kind, self.scale, float(self.m._id_par[tid]),
                                                float(self.m._id_ang[tid]))
            self.cache[tid] = (amp, snr)
        return self.cache[tid]

    def match(self, templates, plan, sync=True):
        w = float(self.opt.get("near_window", 0.0))
        self.w_used = w
        for t in templates:
            tid = int(t.id)
            a64, s64 = self._f64(tid)
            s32 = (s64 * (1.0 + self.noise * _jitter(tid, self.z.shape))).astype(np.float32)
            won = s32 > self.snr
            if w > 0.0:
                big, small = np.maximum(s32, self.snr), np.minimum(s32, self.snr)
                near = (small >= big * np.float32(1.0 - w)) & (self.idx != NONE_ID)
                for c in np.flatnonzero(near):
                    self.events.append((int(c), tid, int(self.idx.flat[c]), float(big.flat[c])))
            self.amp = np.where(won, a64.astype(np.float32), self.amp)
            self.snr = np.where(won, s32, self.snr)
            self.idx = np.where(won, np.uint32(tid), self.idx)

    def snapshot_best(self):
        self.snap = (self.snr.copy(), self.idx.copy())

    def get_best(self):
        return self.amp.copy(), self.snr.copy(), self.idx.copy()

    def set_best(self, amp, snr, idx):
        self.amp, self.snr, self.idx, self.patch = np.array(amp), np.array(snr), np.array(idx), {}

    def rank_candidates(self):
        if getattr(self, "fail", False):                         # (what an overflowed event list answers)
            from scarplet_amd import _lib
            raise _lib.ScarpletHipError("sc_rank_candidates: the event list overflowed")
        keep = 1.0 - self.w_used
        out = []
        for (c, tid, holder, big) in self.events:
            if big >= self.snr.flat[c] * keep:
                out += [(c, tid), (c, holder)]
        s_snr, s_idx = self.snap
        mine = (s_idx != NONE_ID) & (s_idx != self.idx) & (s_snr > 0) & (s_snr >= self.snr * np.float32(keep))
        out += [(int(c), int(s_idx.flat[c])) for c in np.flatnonzero(mine)]
        return np.array(out, dtype=np.uint32).reshape(-1, 2)

    def settle_pairs(self, templates, pairs, n_twin=0, max_work=0.0):
        assert len(templates) == len(self.m._id_par)                 # the descriptors of the WHOLE grid
        lists = {}
        for c, tid in np.asarray(pairs).reshape(-1, 2):
            lists.setdefault(int(c), set()).add(int(tid))
        changed = 0
        for c, ids in lists.items():
            ids.add(int(self.idx.flat[c]))
            best = max(sorted(ids), key=lambda k: (self._f64(k)[1].flat[c], -k))      # float64 argmax, ties to the earlier id
            changed += best != int(self.idx.flat[c])
            a64, s64 = self._f64(best)
            self.idx.flat[c] = best
            self.amp.flat[c], self.snr.flat[c] = a64.flat[c], s64.flat[c]
            self.patch[c] = (a64.flat[c], s64.flat[c])
        return {"flagged_cells": len(lists), "pairs_listed": int(sum(len(v) for v in lists.values())),
                "float64_pairs": 0, "float64_cells": len(lists), "changed_cells": int(changed), "events": len(pairs), "taps": 0}

    def result_array(self):
        out = np.stack(sd.record_planes(self.amp, self.snr, self.idx, self.m._id_par, self.m._id_ang))
        for c, (a, s) in self.patch.items():
            out[0].flat[c], out[3].flat[c] = a, s
        return out


def _exact_orientation_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from torch_transport import TorchTransport
        import scarplet_oracle as orc
        import scarplet_amd as sl
        from scarplet_amd.core import Matcher
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        rng = np.random.default_rng(23)
        z = (np.cumsum(rng.standard_normal((40, 36)), 1) * 0.05 + rng.standard_normal((40, 36)) * 0.03)
        # every orientation twice, 2e-4 rad apart: near-ties within the float32 error - and the chunks (contiguous in grid
        # order) put the two of a pair on DIFFERENT ranks, for 2 ranks and for 3
        ages, angles = [1.0, 4.0, 16.0], np.array([0.3, -0.8, 0.9, 0.3002, -0.8002, 0.9002])
        m = object.__new__(Matcher)
        m.ny, m.nx, m.de, m.core, m.whole = 40, 36, 1.0, (0, 40, 0, 36), True
        noise = 2.5e-4                                       # (below half the 6e-4 window, as the device's errors are)
        m.ctx = _ExactOracleContext(m, z, orc.SCARP, 6, noise)
        m.result_array = lambda: m.ctx.result_array()
        om = sd.OrientationMatcher(rank, world, None, backend="host", transport=TorchTransport(), matcher=m)
        om.search(sl.Scarp, 6, ages, angles, method="fft")           # exact by default: a built-in, a transport
        got = om.result_array()
        # the reference: compare() over float64 maps in fold order (orientation-major ids), ties to the incumbent
        want = np.zeros((4, 40, 36))
        want_id = np.full((40, 36), NONE_ID, np.uint32)
        k = 0
        for ang in angles:
            for age in ages:
                amp, _, _, snr = orc.match_template(z, 1.0, 1.0, orc.SCARP, 6, age, float(ang))
                take = snr > want[3]
                for kk, v in enumerate((amp, age, float(ang), snr)):
                    want[kk] = np.where(take, v, want[kk])
                want_id = np.where(take, np.uint32(k), want_id)
                k += 1
        st = om.exact_stats
        checks = {}
        checks["every cell holds the float64 argmax"] = got is not None and np.array_equal(m.ctx.idx, want_id)
        checks["age and angle planes"] = np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
        settled = np.zeros(40 * 36, bool)
        settled[list(m.ctx.patch)] = True
        settled = settled.reshape(40, 36)
        checks["settled cells carry float64 (amp, snr)"] = np.array_equal(got[3][settled], want[3][settled]) and \
            np.array_equal(got[0][settled], want[0][settled])
        checks["other cells within the float32 error"] = np.allclose(got[3], want[3], rtol=2 * noise, atol=0)
        checks["the settle changed cells"] = st["changed_cells"] > 0
        # the float32 fold alone is NOT the float64 argmax on this surface (the test would prove nothing otherwise)
        m2 = object.__new__(Matcher)
        m2.ny, m2.nx, m2.de, m2.core, m2.whole = 40, 36, 1.0, (0, 40, 0, 36), True
        m2.ctx = _ExactOracleContext(m2, z, orc.SCARP, 6, noise)
        m2.ctx.cache = m.ctx.cache
        om2 = sd.OrientationMatcher(0, 1, None, backend="host", matcher=m2)
        mine2, sp2 = om2.describe(sl.Scarp, 6, ages, angles, method="fft")
        om2.run(mine2, sp2)
        n_off = int((m2.ctx.idx != want_id).sum())
        checks["float32 alone is off in >= 10 cells (%d)" % n_off] = n_off >= 10
        # ... and one rank settling the whole search holds the same record as every rank of the sharded one
        om2.run(mine2, sp2, m2.exact_window_for(om2._keep, sp2), 0)
        checks["one rank's settle = the sharded one"] = np.array_equal(m2.ctx.idx, m.ctx.idx)
        ok = all(checks.values()) or [k_ for k_, v_ in checks.items() if not v_]
        # a rank that cannot list its candidates must not leave the others waiting in the exchange: every rank raises
        from scarplet_amd import _lib
        m.ctx.fail = rank == world - 1
        try:
            om.search(sl.Scarp, 6, ages, angles, method="fft")
            checks["a failed rank fails every rank"] = False
        except _lib.ScarpletHipError as e:
            checks["a failed rank fails every rank"] = "overflowed" in str(e)
        ok = all(checks.values()) or [k_ for k_, v_ in checks.items() if not v_]
        if rank == 0:
            print("exact orientation sharding, %d ranks: %s" % (world, st))
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok))
    except Exception as e:                       # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_exact_orientation_sharding_settles_cross_rank_ties_over_gloo(world):
    """The N > 1 path of exact mode for the orientation sharding (dist.OrientationMatcher.run), on CPU ranks over gloo: the
    records are folded through the transport, every rank lists its candidates against the FOLDED record, the lists are
    exchanged and every rank settles the union - every rank ends with the float64 argmax of the whole search in every cell
    (every orientation has a near twin on ANOTHER rank: near-ties no rank's event list holds)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exact_orientation_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)], res


def test_fold_records_is_the_rule_of_sc_fold_ranks():
    rng = np.random.default_rng(5)
    parts = []
    for r in range(3):
        snr = rng.integers(0, 4, size=(5, 6)).astype(np.float32)
        parts.append((snr + 10 * r, snr, rng.permutation(30).reshape(5, 6).astype(np.uint32) + 30 * (2 - r)))
    parts[1][1][2, 2] = np.nan
    amp, snr, idx = sd.fold_records(parts)
    for i in range(5):
        for j in range(6):
            if (i, j) == (2, 2):
                assert np.isnan(snr[i, j]) and idx[i, j] == parts[1][2][i, j]
                continue
            r = max(range(3), key=lambda r_: (parts[r_][1][i, j], -int(parts[r_][2][i, j])))
            assert snr[i, j] == parts[r][1][i, j] and idx[i, j] == parts[r][2][i, j] and amp[i, j] == parts[r][0][i, j]


def test_package_imports_no_process_group_library():
    """north_star: host code is Python over ctypes, no PyTorch in the product."""
    import subprocess
    code = ("import sys; import scarplet_amd, scarplet_amd.dist, scarplet_amd.core; "
            "assert 'torch' not in sys.modules, 'torch imported by the package'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scarplet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import torch" not in src, f


def test_halo_for_search_covers_the_reach():
    # output cell i reads curvature rows i - pmax + oy .. i - pmin + oy, +1 for the stencil
    assert sd.halo_for_search((-10, 9, -5, 6), 100, 101) == (10, 11, 6, 7)
    assert sd.halo_for_search((-10, 9, -5, 6), 101, 100) == (9, 12, 7, 6)


def test_orientation_chunks_and_host_fold():
    for (n, r) in [(181, 8), (35, 2), (3, 5), (91, 4)]:
        ch = sd.orientation_chunks(n, r)
        assert len(ch) == r and ch[0][0] == 0 and ch[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        sizes = [b - a for a, b in ch]
        assert max(sizes) - min(sizes) <= 1
    # fold_host: greatest SNR wins, ties keep the earlier rank, NaN sticks
    rng = np.random.default_rng(3)
    parts = []
    for r in range(4):
        snr = rng.integers(0, 5, size=(6, 7)).astype(float)          # many exact ties
        parts.append(np.stack([snr + 10 * (r + 1), np.full((6, 7), r + 1.0), np.full((6, 7), -r - 1.0), snr]))
    parts[2][3, 1, 1] = np.nan
    got = sd.fold_host(parts)
    for i in range(6):
        for j in range(7):
            col = [p[3, i, j] for p in parts]
            if (i, j) == (1, 1):
                first_nan = 2
                before = max(range(2), key=lambda r: (col[r], -r))
                want = first_nan if not np.isnan(col[before]) else before
                assert np.isnan(got[3, i, j]) and got[1, i, j] == want + 1
                continue
            want = max(range(4), key=lambda r: (col[r], -r))            # max SNR, then the earliest rank
            assert got[3, i, j] == col[want] and got[1, i, j] == want + 1 and got[0, i, j] == col[want] + 10 * (want + 1)


# ---- the final gather ships the float32 record, the root converts it -------------------------
def test_record_planes_is_the_device_conversion():
    """dist.record_planes = k_result (sc_api.hip): amp and snr widened, age / angle looked up by id,
    zero where no template has won (SC_ID_NONE)."""
    rng = np.random.default_rng(2)
    par, ang = np.array([1.0, 10.0, 100.0, 1.0, 10.0, 100.0]), np.array([-0.5, -0.5, -0.5, 0.5, 0.5, 0.5])
    idx = rng.integers(0, 6, size=(5, 7)).astype(np.uint32)
    idx[0, 0] = idx[4, 6] = 0xFFFFFFFF
    amp = rng.standard_normal((5, 7)).astype(np.float32)
    snr = rng.random((5, 7)).astype(np.float32)
    a, g, o, s = sd.record_planes(amp, snr, idx, par, ang)
    assert a.dtype == g.dtype == o.dtype == s.dtype == np.float64
    assert np.array_equal(a, amp.astype(np.float64)) and np.array_equal(s, snr.astype(np.float64))
    won = idx != 0xFFFFFFFF
    assert np.array_equal(g[won], par[idx[won]]) and np.array_equal(o[won], ang[idx[won]])
    assert (g[~won] == 0).all() and (o[~won] == 0).all()
    assert sd.RECORD_BYTES == amp.itemsize + snr.itemsize + idx.itemsize


class _RecordContext(object):
    """Stands in for a rank's GPU context in the gather test: a record over the rank's core."""

    def __init__(self, core, seed):
        h, w = core[1] - core[0], core[3] - core[2]
        rng = np.random.default_rng(seed)
        self.rec = (rng.standard_normal((h, w)).astype(np.float32), rng.random((h, w)).astype(np.float32),
                    rng.integers(0, 7, size=(h, w)).astype(np.uint32))     # id 6: beyond the 6-entry table

    def get_best(self):
        return self.rec


def _gather_worker(rank, world, port, q):
    try:
        import types
        import torch.distributed as dist
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from torch_transport import TorchTransport
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
        ny, nx = 37, 41
        dm = object.__new__(sd.DistMatcher)                  # host side only: no device behind it
        dm.rank, dm.nranks, dm.ny, dm.nx, dm.dx, dm.dy = rank, world, ny, nx, 1.0, 1.0
        dm.backend, dm.transport, dm.cores = "host", TorchTransport(), None
        dm.py, dm.px = sd.grid_dims(world, ny, nx)
        params, angles = np.array([1.0, 10.0, 100.0]), np.array([-0.5, 0.5])
        dm.m = types.SimpleNamespace(ctx=_RecordContext(dm.core(), 100 + rank), params=params, angles=angles)
        sent = []
        real_gather = dm.transport.gather

        def spy(obj, dst):
            sent.append(obj)
            return real_gather(obj, dst)
        dm.transport.gather = spy
        got = dm.gather(0)
        # what travelled: this rank's core and three 4-byte planes of exactly its cells - 12 B per cell
        core, rec = sent[0]
        cells = (core[1] - core[0]) * (core[3] - core[2])
        ok = sum(a.nbytes for a in rec) == sd.RECORD_BYTES * cells and [a.dtype.itemsize for a in rec] == [4, 4, 4]
        if rank == 0:
            par, ang = np.repeat(params, len(angles)), np.tile(angles, len(params))
            lay = sd.Layout(ny, nx, dm.py, dm.px, (0, 0, 0, 0))
            want = np.zeros((4, ny, nx))
            for r in range(world):
                c = lay.core(r)
                rec_r = _RecordContext(c, 100 + r).rec
                want[:, c[0]:c[1], c[2]:c[3]] = np.stack(sd.record_planes(rec_r[0], rec_r[1], rec_r[2], par, ang))
            ok = ok and np.array_equal(np.stack(got), want)
        else:
            ok = ok and got is None
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(ok)))
    except Exception:                            # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 4])
def test_gather_ships_the_record_over_gloo(world):
    """DistMatcher.gather on the host transport mirrors sc_gather_result's layout: 12 bytes per cell per
    rank (amp f32, snr f32, id u32), converted to the four float64 planes at the root."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)], res
