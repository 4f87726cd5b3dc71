"""Multi-GPU search, one process per GPU.  Two ways to shard:

* by ORIENTATION (``OrientationMatcher``, at the end of this module): every rank holds the whole
  DEM, searches a chunk of the orientation grid and the running-best records are folded over
  RCCL - the reference's own pool-over-orientations + compare() (core.py:180-195), bit-identical
  to the single-GPU search.  The way to go whenever the DEM fits one GPU;
* by SPACE (``DistMatcher``), for DEMs that do not: the DEM cut into rectangles, described next.


The reference has no distributed path (its only parallelism is a process pool
over orientations, core.py:180-183); its authors tiled large DEMs by hand on
a cluster (CHANGELOG.md:9).  Here a search over a DEM too large - or too slow -
for one GPU is sharded spatially: every rank owns a rectangle of output cells,
needs the elevations within one template half-extent (+1 cell for the
curvature stencil) around it, and gets that halo from its neighbours.  The
reference's convolution is circular, so the tile grid is a torus: the halo of
an edge tile comes from the opposite side of the DEM.

Nothing else is exchanged: results are disjoint tiles, so the "argmax
all-reduce" of a replicated design degenerates to a gather of tiles.

This module is pure host logic plus two executors for the same transfer list:
  * ``exchange_host`` - point-to-point on host arrays through a caller-supplied
    *transport* (CPU tests and bring-up on machines without xGMI);
  * ``Context.halo_exchange`` (sc_halo_exchange) - grouped ncclSend/ncclRecv
    on device buffers over RCCL.

The package itself depends on no process-group library.  Whoever launches the
ranks (bench.py, the tests, a user's MPI or torchrun script) passes a transport:

    class Transport:                      # duck-typed
        def broadcast_bytes(self, payload):     # rank 0's bytes -> every rank
        def exchange(self, sends, recvs):       # sends: [(peer, tag, ndarray)],
                                                # recvs: [(peer, tag, shape)]
                                                # -> list of received float64 arrays
        def gather(self, obj, dst):             # -> list of objects on dst, else None

The RCCL path needs ``broadcast_bytes`` only (the 128-byte communicator id);
tools/torch_transport.py implements all three over torch.distributed.
"""

import numpy as np

from scarplet_amd import _lib, _plan


def grid_dims(nranks, ny, nx):
    """py x px process grid with tiles as square as possible."""
    best = None
    for py in range(1, nranks + 1):
        if nranks % py:
            continue
        px = nranks // py
        aspect = abs(np.log((ny / py) / (nx / px)))
        if best is None or aspect < best[0]:
            best = (aspect, py, px)
    return best[1], best[2]


def cuts(n, parts):
    """parts+1 cut positions splitting n cells as evenly as possible."""
    return np.array([(n * k) // parts for k in range(parts + 1)], dtype=int)


def grid_cores(ny, nx, py, px):
    """Cores (y0, y1, x0, x1) of an even py x px grid, rank = ry*px + rx."""
    cy, cx = cuts(ny, py), cuts(nx, px)
    return [(int(cy[r // px]), int(cy[r // px + 1]), int(cx[r % px]), int(cx[r % px + 1]))
            for r in range(py * px)]


def tile_partition(nranks, nty, ntx, budget=200000):
    """Cut an nty x ntx grid of FFT tiles into ``nranks`` rectangles of as few tiles as possible
    each (the cost of a rank's search is its tile count, not its area: an even 2x4 split of the
    10000^2 benchmark DEM gives every rank 3x2 tiles of 2048^2 where 36 / 8 = 4.5 would do).
    Among the partitions with the smallest maximum the one with the shortest total boundary
    (least halo) found within ``budget`` search nodes is returned, as
    [(ty0, ty1, tx0, tx1)] in tile units - or None when there are fewer tiles than ranks.

    Exhaustive placement: the first free cell in row-major order is the top-left corner of
    the next rectangle.  6x6 tiles over 8 ranks: four 1x5 strips around four 2x2 blocks."""
    cells = nty * ntx
    if nranks < 1 or cells < nranks:
        return None
    for cap in range(-(-cells // nranks), cells + 1):
        shapes = sorted(((h, w) for h in range(1, nty + 1) for w in range(1, ntx + 1) if h * w <= cap),
                        key=lambda hw: (-hw[0] * hw[1], abs(hw[0] - hw[1])))
        used = np.zeros((nty, ntx), dtype=bool)
        best = [None, None]
        nodes = [0]
        cur = []

        def rec(k_left, free, perim):
            if nodes[0] > budget:
                return
            nodes[0] += 1
            if free == 0:
                if k_left == 0 and (best[0] is None or perim < best[0]):
                    best[0], best[1] = perim, list(cur)
                return
            if k_left == 0 or free > k_left * cap or free < k_left:
                return
            if best[0] is not None and perim >= best[0]:
                return
            i = int(np.argmin(used.ravel()))           # first free cell
            y, x = divmod(i, ntx)
            for (h, w) in shapes:
                if y + h > nty or x + w > ntx or used[y:y + h, x:x + w].any():
                    continue
                used[y:y + h, x:x + w] = True
                cur.append((y, y + h, x, x + w))
                rec(k_left - 1, free - h * w, perim + h + w)
                cur.pop()
                used[y:y + h, x:x + w] = False

        rec(nranks, cells, 0)
        if best[1] is not None:
            return best[1]
    return None


def tile_cores(nranks, ny, nx, bbox):
    """Cores for ``nranks`` ranks aligned with the overlap-save tiles a template batch with
    support box ``bbox`` gets (_plan.choose_tile), or None when that is no better than the
    even grid of grid_dims (fewest tiles on the busiest rank decides)."""
    pmin, pmax, qmin, qmax = bbox
    try:
        _, vy, nty, _ = _plan.choose_tile(ny, pmax - pmin, ny, False)
        _, vx, ntx, _ = _plan.choose_tile(nx, qmax - qmin, nx, False)
    except ValueError:
        return None

    def tiles_of(core):
        return _plan.choose_tile(core[1] - core[0], pmax - pmin, ny, False)[2] * \
            _plan.choose_tile(core[3] - core[2], qmax - qmin, nx, False)[2]

    py, px = grid_dims(nranks, ny, nx)
    even = max(tiles_of(c) for c in grid_cores(ny, nx, py, px))
    part = tile_partition(nranks, nty, ntx)
    if part is None or max((a1 - a0) * (b1 - b0) for (a0, a1, b0, b1) in part) >= even:
        return None
    return [(min(a0 * vy, ny), min(a1 * vy, ny), min(b0 * vx, nx), min(b1 * vx, nx))
            for (a0, a1, b0, b1) in part]


class Layout(object):
    """Cores of a ny x nx DEM over the ranks - an even py x px grid (rank = ry*px + rx), or
    any list ``cores`` of rectangles that tile the DEM (tile_cores) - and the halo around them."""

    def __init__(self, ny, nx, py, px, halo, cores=None):
        self.ny, self.nx, self.py, self.px = ny, nx, py, px
        self.cores = [tuple(int(v) for v in c) for c in cores] if cores is not None \
            else grid_cores(ny, nx, py, px)
        area = 0
        for k, (y0, y1, x0, x1) in enumerate(self.cores):
            if not (0 <= y0 < y1 <= ny and 0 <= x0 < x1 <= nx):
                raise ValueError("core outside the DEM")
            area += (y1 - y0) * (x1 - x0)
            for (v0, v1, u0, u1) in self.cores[:k]:
                if y0 < v1 and v0 < y1 and x0 < u1 and u0 < x1:
                    raise ValueError("cores overlap")
        if area != ny * nx:
            raise ValueError("cores must tile the DEM exactly")
        self.halo = tuple(int(h) for h in halo)       # (y_lo, y_hi, x_lo, x_hi)
        if max(self.halo[:2]) >= ny or max(self.halo[2:]) >= nx:
            raise ValueError("halo larger than the DEM")

    @property
    def nranks(self):
        return len(self.cores)

    def core(self, rank):
        return self.cores[rank]

    def block_origin(self, rank):
        c = self.core(rank)
        return c[0] - self.halo[0], c[2] - self.halo[2]

    def block_shape(self, rank):
        c = self.core(rank)
        return (c[1] - c[0] + self.halo[0] + self.halo[1],
                c[3] - c[2] + self.halo[2] + self.halo[3])

    def transfers(self):
        """Global, ordered list of rectangles that fill every rank's halo:
        (src, dst, sy0, sx0, dy0, dx0, h, w) with (sy0, sx0) in the SOURCE
        rank's block coordinates and (dy0, dx0) in the destination's.  Every
        rank derives the same list, which is what keeps sends and receives
        between a pair in matching order."""
        out = []
        for dst in range(self.nranks):
            oy, ox = self.block_origin(dst)
            bh, bw = self.block_shape(dst)
            for ky in (-1, 0, 1):
                for kx in (-1, 0, 1):
                    for src in range(self.nranks):
                        if src == dst and ky == 0 and kx == 0:
                            continue
                        sc = self.core(src)
                        # source core as seen in the destination's unwrapped
                        # coordinates (periodic image ky, kx)
                        y0 = max(sc[0] + ky * self.ny, oy)
                        y1 = min(sc[1] + ky * self.ny, oy + bh)
                        x0 = max(sc[2] + kx * self.nx, ox)
                        x1 = min(sc[3] + kx * self.nx, ox + bw)
                        if y1 <= y0 or x1 <= x0:
                            continue
                        soy, sox = self.block_origin(src)
                        out.append((src, dst,
                                    y0 - ky * self.ny - soy, x0 - kx * self.nx - sox,
                                    y0 - oy, x0 - ox, y1 - y0, x1 - x0))
        return out

    def rank_transfers(self, rank, with_tag=False):
        """This rank's view: sc_xfer tuples (peer, kind, sy0, sx0, dy0, dx0,
        h, w) in global order (``with_tag`` appends the global index, used as
        the message tag by the host executor)."""
        out = []
        for tag, (src, dst, sy0, sx0, dy0, dx0, h, w) in enumerate(self.transfers()):
            if src == rank and dst == rank:
                x = (rank, _lib.XFER_LOCAL, sy0, sx0, dy0, dx0, h, w)
            elif dst == rank:
                x = (src, _lib.XFER_RECV, 0, 0, dy0, dx0, h, w)
            elif src == rank:
                x = (dst, _lib.XFER_SEND, sy0, sx0, 0, 0, h, w)
            else:
                continue
            out.append(x + (tag,) if with_tag else x)
        return out


def halo_for_search(bbox, ny, nx):
    """Halo a rank needs for a template batch with support box ``bbox``:
    the template reach (_plan.halo_for) plus one cell for the curvature
    stencil (dem.py:88-101)."""
    return tuple(h + 1 for h in _plan.halo_for(bbox, ny, nx))


def assemble_block_reference(z, layout, rank):
    """What a rank's block must contain: the DEM indexed modulo its size.
    (Used by the tests as the ground truth of an exchange.)"""
    oy, ox = layout.block_origin(rank)
    bh, bw = layout.block_shape(rank)
    ii = (oy + np.arange(bh)) % layout.ny
    jj = (ox + np.arange(bw)) % layout.nx
    return np.asarray(z)[np.ix_(ii, jj)]


def exchange_host(core, layout, rank, transport):
    """Host executor of the transfer list: this rank's sends and receives go
    through ``transport.exchange`` (see the module docstring), periodic images
    of the rank's own core are copied in place."""
    bh, bw = layout.block_shape(rank)
    hy, _, hx, _ = layout.halo
    blk = np.zeros((bh, bw), dtype=np.float64)
    c = layout.core(rank)
    blk[hy:hy + c[1] - c[0], hx:hx + c[3] - c[2]] = core
    mine = layout.rank_transfers(rank, with_tag=True)
    sends = [(peer, tag, np.ascontiguousarray(blk[sy0:sy0 + h, sx0:sx0 + w]))
             for (peer, kind, sy0, sx0, dy0, dx0, h, w, tag) in mine if kind == _lib.XFER_SEND]
    rinfo = [(peer, tag, dy0, dx0, h, w)
             for (peer, kind, sy0, sx0, dy0, dx0, h, w, tag) in mine if kind == _lib.XFER_RECV]
    got = transport.exchange(sends, [(peer, tag, (h, w)) for (peer, tag, _, _, h, w) in rinfo]) \
        if (sends or rinfo) else []
    for (peer, tag, dy0, dx0, h, w), arr in zip(rinfo, got):
        blk[dy0:dy0 + h, dx0:dx0 + w] = arr
    for (peer, kind, sy0, sx0, dy0, dx0, h, w, tag) in mine:
        if kind == _lib.XFER_LOCAL:
            blk[dy0:dy0 + h, dx0:dx0 + w] = blk[sy0:sy0 + h, sx0:sx0 + w]
    return blk


class DistMatcher(object):
    """Per-rank driver of a tiled search.

    ``z_core`` is this rank's rectangle of the DEM (layout.core(rank)).
    ``backend`` 'rccl' exchanges halos on the GPUs (needs
    ``transport.broadcast_bytes`` - or the bare ``broadcast_bytes`` callable -
    for the communicator id when nranks > 1); 'host' moves host arrays through
    ``transport`` ('gloo' is accepted as an alias of 'host')."""

    def __init__(self, rank, nranks, shape, dx, dy, device=0, backend="rccl",
                 broadcast_bytes=None, transport=None):
        from scarplet_amd.core import Matcher
        self.rank, self.nranks = rank, nranks
        self.ny, self.nx = shape
        self.dx, self.dy = dx, dy
        self.backend = "host" if backend == "gloo" else backend
        backend = self.backend
        self.transport = transport
        if backend not in ("rccl", "host"):
            raise ValueError("backend must be 'rccl' or 'host'")
        if backend == "host" and nranks > 1 and transport is None:
            raise ValueError("the host backend needs a transport (scarplet_amd/dist.py docstring)")
        self.py, self.px = grid_dims(nranks, self.ny, self.nx)
        self.cores = None                     # None: the even py x px grid; else partition_for's
        self.m = Matcher(device=device)
        # describe() only needs the grid geometry
        self.m.ny, self.m.nx, self.m.de = self.ny, self.nx, dx
        if backend == "rccl" and nranks > 1:
            if broadcast_bytes is None:
                if transport is None:
                    raise ValueError("RCCL with more than one rank needs broadcast_bytes= or "
                                     "transport= to share the communicator id")
                broadcast_bytes = transport.broadcast_bytes
            uid = self.m.ctx.comm_unique_id() if rank == 0 else None
            uid = broadcast_bytes(uid)
            self.m.ctx.comm_init(uid, rank, nranks)

    def partition_for(self, bbox):
        """Choose the ranks' cores for a template batch with support box ``bbox``: rectangles
        of whole FFT tiles when that leaves the busiest rank fewer tiles than the even grid
        (tile_cores), else the grid.  Every rank must call it with the same box, before
        ``core()``; returns this rank's core."""
        self.cores = tile_cores(self.nranks, self.ny, self.nx, bbox)
        return self.core()

    def prepare(self, Template, scale, params, angles, **kwargs):
        """``partition_for`` the support box of a search (same arguments as ``search``)."""
        params = np.atleast_1d(np.asarray(params, dtype=float))
        angles = np.atleast_1d(np.asarray(angles, dtype=float))
        _, bbox, _ = self.m.describe(Template, scale, params, angles, **kwargs)
        return self.partition_for(bbox)

    def _layout(self, halo):
        return Layout(self.ny, self.nx, self.py, self.px, halo, cores=self.cores)

    def core(self):
        return self._layout((0, 0, 0, 0)).core(self.rank)

    def load(self, z_core, bbox):
        """Exchange halos sized for a template batch and hand the block to
        the GPU."""
        halo = halo_for_search(bbox, self.ny, self.nx)
        self.layout = self._layout(halo)
        core = self.layout.core(self.rank)
        origin = self.layout.block_origin(self.rank)
        bshape = self.layout.block_shape(self.rank)
        z_core = np.ascontiguousarray(z_core, dtype=np.float64)
        assert z_core.shape == (core[1] - core[0], core[3] - core[2])
        if self.backend == "rccl":
            z_dev = self.m.ctx.halo_exchange(z_core, halo,
                                             self.layout.rank_transfers(self.rank))
            self.m.set_block(z_dev, origin, (self.ny, self.nx), core, self.dx,
                             self.dy, block_shape=bshape)
        else:
            blk = exchange_host(z_core, self.layout, self.rank, self.transport)
            self.m.set_block(blk, origin, (self.ny, self.nx), core, self.dx, self.dy)
        return self

    def search(self, Template, scale, params, angles, z_core, method="fft",
               group=None, exact=None, **kwargs):
        """This rank's part of the tiled search.  ``exact`` (default: on for the built-in template classes, as in
        ``scarplet_amd.match``): the near-ties of the rank's own block are settled in float64 on its device before the
        gather (sc_settle_exact: the halo covers the templates' reach and the curvature stencil) - the record that
        travels carries the float64 argmax, its amplitude and SNR rounded to float32."""
        params = np.atleast_1d(np.asarray(params, dtype=float))
        angles = np.atleast_1d(np.asarray(angles, dtype=float))
        arr, bbox, max_area = self.m.describe(Template, scale, params, angles, **kwargs)
        self.load(z_core, bbox)
        self.m.plan, sp = self.m.plan_for(bbox, max_area, method, group,
                                          n_params=len(params))
        if exact is None:
            exact = all(int(arr[k].kind) != 2 for k in (0, len(arr) - 1))         # (2: SC_KIND_WINDOW, a host-uploaded plugin)
        self.exact_stats = self.m.run_described(arr, sp, self.m.exact_window_for(arr, sp) if exact else 0.0,
                                                self.m.end_twins(arr, len(params), angles))
        self.m.params, self.m.angles = params, angles
        return self

    def result(self):
        """This rank's tile of (amp, age, angle, snr)."""
        return self.m.result()

    def gather(self, dst=0, out=None):
        """Assemble the full maps on rank ``dst`` (None elsewhere): over RCCL
        (sc_gather_result, device to root's host array) with the 'rccl' backend,
        through ``transport.gather`` with the host backend.  ``out``: a (4, ny, nx)
        float64 array to fill on rank ``dst`` (a search repeated on the same DEM need
        not fault in 32 bytes per cell of fresh memory every time)."""
        if self.backend == "rccl":
            lay = self._layout((0, 0, 0, 0))
            cores = [lay.core(r) for r in range(self.nranks)]
            m = self.m
            out = m.ctx.gather_result(dst, cores, (self.ny, self.nx),
                                      np.repeat(m.params, len(m.angles)),
                                      np.tile(m.angles, len(m.params)), self.rank == dst, out=out)
            return tuple(out) if self.rank == dst else None
        # host transport: the same layout as sc_gather_result - every rank ships the float32 RECORD of its
        # core (amp, snr, id: 12 B per cell), the root turns each into the four float64 planes with the id
        # tables every rank shares (record_planes) and places it
        m = self.m
        rec = m.ctx.get_best()
        tiles = [(self.core(), rec)] if self.nranks == 1 else self.transport.gather((self.core(), rec), dst)
        if self.rank != dst:
            return None
        par, ang = np.repeat(m.params, len(m.angles)), np.tile(m.angles, len(m.params))
        out = [np.zeros((self.ny, self.nx)) for _ in range(4)] if out is None else list(out)
        for core, (amp, snr, idx) in tiles:
            res = record_planes(amp, snr, idx, par, ang)
            for k in range(4):
                out[k][core[0]:core[1], core[2]:core[3]] = res[k]
        return tuple(out)


RECORD_BYTES = 12       # per cell between ranks: amp float32, snr float32, template id uint32 (sc_internal.h)


def record_planes(amp, snr, idx, param_of_id, angle_of_id):
    """The running-best record -> the reference's four float64 planes (amp, age, angle, snr;
    core.py:190-195), as k_result does on the device: a cell no template has won (id beyond the
    table, SC_ID_NONE) has age and angle 0."""
    amp, snr, idx = np.asarray(amp), np.asarray(snr), np.asarray(idx)
    par = np.asarray(param_of_id, dtype=np.float64)
    ang = np.asarray(angle_of_id, dtype=np.float64)
    won = idx < len(par)
    safe = np.where(won, idx, 0)
    return (amp.astype(np.float64), np.where(won, par[safe], 0.0), np.where(won, ang[safe], 0.0),
            snr.astype(np.float64))


# ---- orientation sharding -------------------------------------------------------------------
def orientation_chunks(n_angles, nranks):
    """[b0, b1) of the orientation grid for every rank: contiguous, in grid order, as even as
    possible (a rank may get none when there are more ranks than orientations)."""
    c = cuts(n_angles, nranks)
    return [(int(c[r]), int(c[r + 1])) for r in range(nranks)]


def fold_host(results):
    """compare() (core.py:198-243) over the ranks' (4, ny, nx) records in rank order, started
    from the first: a later rank replaces a cell only with a strictly greater SNR, which is
    the smaller-id rule of sc_fold_ranks for ranks holding increasing orientation chunks.
    NaN SNRs stick, as in sc_match."""
    best = np.array(results[0], dtype=np.float64, copy=True)
    for r in results[1:]:
        r = np.asarray(r)
        with np.errstate(invalid="ignore"):
            take = (r[3] > best[3]) | (np.isnan(r[3]) & ~np.isnan(best[3]))
        best[:, take] = r[:, take]
    return best


def fold_records(parts):
    """sc_fold_ranks on host records [(amp float32, snr float32, id uint32)]: per cell the greatest SNR, equal SNRs to the
    smaller id (the earlier template of the fold order), a NaN SNR beats every number; the amplitude follows the winner."""
    amp, snr, idx = [np.array(a, copy=True) for a in parts[0]]
    for (a, s, i) in parts[1:]:
        with np.errstate(invalid="ignore"):
            take = (s > snr) | ((s == snr) & (i < idx)) | (np.isnan(s) & ~np.isnan(snr))
        amp[take], snr[take], idx[take] = a[take], s[take], i[take]
    return amp, snr, idx


class OrientationMatcher(object):
    """Per-rank driver of an orientation-sharded search - the reference's own parallelism
    (a pool over orientations, core.py:180-183, folded by compare()) across GPUs.

    Every rank holds the WHOLE DEM and searches a contiguous chunk of the orientation grid
    with all ages; the ranks' running-best records are then folded into one
    (``sc_fold_ranks``: two all-reduces over RCCL/xGMI, or ``fold_host`` through the
    transport with the host backend).  The templates carry global ids in fold order
    (orientation-major) and every rank plans for the support box of the WHOLE grid, so the
    folded record is, bit for bit, the one a single context searching all orientations holds.
    Use it whenever the DEM fits one GPU: no tile quantisation, no replicated template
    transforms - the spatial DistMatcher is for DEMs that do not fit.

    ``data`` is a DEMGrid (or anything Matcher accepts)."""

    def __init__(self, rank, nranks, data, device=0, backend="rccl", broadcast_bytes=None,
                 transport=None, matcher=None):
        from scarplet_amd.core import Matcher
        self.rank, self.nranks = rank, nranks
        self.backend = "host" if backend == "gloo" else backend
        self.transport = transport
        if self.backend not in ("rccl", "host"):
            raise ValueError("backend must be 'rccl' or 'host'")
        if self.backend == "host" and nranks > 1 and transport is None:
            raise ValueError("the host backend needs a transport (scarplet_amd/dist.py docstring)")
        # (matcher=: a ready Matcher-like object instead of Matcher(data) - the CPU tests of the
        #  multi-rank logic hand in one whose context answers from the oracle)
        self.m = matcher if matcher is not None else Matcher(data, device=device)
        if self.backend == "rccl" and nranks > 1:
            if broadcast_bytes is None:
                if transport is None:
                    raise ValueError("RCCL with more than one rank needs broadcast_bytes= or "
                                     "transport= to share the communicator id")
                broadcast_bytes = transport.broadcast_bytes
            uid = self.m.ctx.comm_unique_id() if rank == 0 else None
            uid = broadcast_bytes(uid)
            self.m.ctx.comm_init(uid, rank, nranks)

    def describe(self, Template, scale, params, angles, method="auto", group=None, **kwargs):
        """(descriptors of this rank's chunk, plan struct): the whole grid is described - ids
        ib * n_params + ia, the fold order - and planned for; the chunk is a slice of it."""
        m = self.m
        params = np.atleast_1d(np.asarray(params, dtype=float))
        angles = np.atleast_1d(np.asarray(angles, dtype=float))
        n_par = len(params)
        arr, bbox, area = m.describe(Template, scale, params, angles,
                                     id_of=lambda ia, ib: ib * n_par + ia, **kwargs)
        m.plan, sp = m.plan_for(bbox, area, method, group, n_params=n_par)
        b0, b1 = orientation_chunks(len(angles), self.nranks)[self.rank]
        n = (b1 - b0) * n_par
        mine = (_lib.sc_template * n).from_buffer(arr, b0 * n_par * _lib.C.sizeof(_lib.sc_template)) \
            if n else None
        m._id_par, m._id_ang = np.tile(params, len(angles)), np.repeat(angles, n_par)
        m.params, m.angles = params, angles
        self._keep = arr                      # the slice borrows its memory
        return mine, sp

    def run(self, mine, sp, exact_window=0.0, n_twin=0):
        """One search step: reset, this rank's templates, fold over the ranks - and, with ``exact_window`` > 0, the
        float64 settle of the near-ties of the WHOLE search (``n_twin``: Matcher.end_twins of the whole list):

        every rank matches with the near-tie window on and keeps its own record's (snr, id) (sc_snapshot_best); after
        the fold it lists what it knows to lie within the window of the FOLDED record - the templates of its events and
        its own holder (sc_rank_candidates: a near-tie between templates of two ranks is in no rank's event list, but
        each rank knows its side of it); the lists travel between the devices (sc_exchange_candidates: two all-gathers over
        RCCL) or, host backend, through the transport (a few megabytes; rank order, so every rank holds the same union) and every rank scores the union in float64 with the descriptors of the whole grid
        (sc_settle_pairs).  Same pairs, same arithmetic: the records agree bit for bit on every rank without a further
        collective, and carry the reference's float64 argmax (compare(), core.py:230-240)."""
        import time
        ctx = self.m.ctx
        exact = exact_window > 0.0
        if exact and self.nranks > 1 and self.backend == "host" and self.transport is None:
            raise ValueError("exact mode over %d ranks needs transport= (the ranks exchange their candidate lists)" % self.nranks)
        ctx.reset_best()
        if exact:
            ctx.set_option("near_window", float(exact_window))
        try:
            if mine is not None:
                ctx.match(mine, sp)
        finally:
            if exact:
                ctx.set_option("near_window", 0.0)
        if exact:
            ctx.snapshot_best()
        self._folded = None
        t0 = time.perf_counter()
        if self.backend == "rccl":
            ctx.fold_ranks()
        elif self.nranks > 1 and exact:
            # the raw records (12 bytes per cell), folded by sc_fold_ranks' rule on rank 0, back onto every device
            parts = self.transport.gather(ctx.get_best(), 0)
            rec = fold_records(parts) if self.rank == 0 else None
            blob = self.transport.broadcast_bytes(b"".join(a.tobytes() for a in rec) if self.rank == 0 else None)
            h, w = ctx.core_shape()
            n = 4 * h * w
            ctx.set_best(np.frombuffer(blob, np.float32, h * w, 0).reshape(h, w),
                         np.frombuffer(blob, np.float32, h * w, n).reshape(h, w),
                         np.frombuffer(blob, np.uint32, h * w, 2 * n).reshape(h, w))
        elif self.nranks > 1:
            parts = self.transport.gather(self.m.result_array(), 0)
            self._folded = fold_host(parts) if self.rank == 0 else None
        self.exact_stats = None
        if exact:
            # (a rank that cannot list its candidates - an overflowed event list - must not leave the others waiting in the
            #  exchange: it takes part with a marker, and every rank raises the same error)
            failed = None
            if self.backend == "rccl":
                # on the devices: the list stays where sc_rank_candidates wrote it, two all-gathers make it every rank's
                try:
                    ctx.rank_candidates(fetch=False)
                except _lib.ScarpletHipError as e:
                    failed = e
                ctx.exchange_candidates()                 # (raises on every rank when some rank has no list)
                if failed is not None:
                    raise failed
                pairs = None
            else:
                try:
                    pairs = ctx.rank_candidates()        # (a rank without templates: no events, nothing held - empty)
                except _lib.ScarpletHipError as e:
                    failed, pairs = e, None
            if self.nranks > 1 and self.backend != "rccl":
                parts = self.transport.gather(pairs if failed is None else "failed: %s" % failed, 0)
                if self.rank == 0:
                    bad = [p_ for p_ in parts if isinstance(p_, str)]
                    blob = (b"!" + bad[0].encode()) if bad else (b"=" + np.concatenate(parts).tobytes())
                else:
                    blob = None
                blob = self.transport.broadcast_bytes(blob)
                if blob[:1] == b"!":
                    raise _lib.ScarpletHipError("exact mode: a rank could not list its candidates (%s)" % blob[1:].decode())
                pairs = np.frombuffer(blob, np.uint32, offset=1).reshape(-1, 2)
            elif failed is not None:
                raise failed
            self.exact_stats = ctx.settle_pairs(self._keep, pairs, n_twin, self.m.EXACT_MAX_F64)
            self.exact_stats["route"] = "device, %d ranks' candidates" % self.nranks
        # wall time of the fold (and, exact mode, of the candidates' exchange and settle) on this rank, the wait for slower
        # ranks included (bench.py: fold_ms)
        self.fold_seconds = getattr(self, "fold_seconds", 0.0) + (time.perf_counter() - t0)
        self._exact_run = exact

    def search(self, Template, scale, params, angles, method="auto", group=None, exact=None, **kwargs):
        """``exact`` (default: on for the built-in template classes wherever the ranks can exchange their candidate
        lists - one rank, RCCL, or a transport): the float64 argmax of the whole search, as ``scarplet_amd.match`` delivers."""
        if getattr(self.m, "nan_dem", False):      # the reference's all-NaN maps, on every rank
            self.m.search(Template, scale, params, angles, method=method, **kwargs)
            self._nan = True
            return self
        self._nan = False
        mine, sp = self.describe(Template, scale, params, angles, method, group, **kwargs)
        arr = self._keep
        if exact is None:
            exact = all(int(arr[k].kind) != 2 for k in (0, len(arr) - 1)) and \
                (self.nranks == 1 or self.backend == "rccl" or self.transport is not None)
        self.run(mine, sp, self.m.exact_window_for(arr, sp) if exact else 0.0,
                 self.m.end_twins(arr, len(self.m.params), self.m.angles))
        return self

    def result_array(self):
        """(4, ny, nx): amp, age, angle, snr - on every rank with the 'rccl' backend (the fold is
        an all-reduce), on rank 0 only (None elsewhere) with the host backend."""
        if self.backend == "rccl" or self.nranks == 1 or getattr(self, "_nan", False) or getattr(self, "_exact_run", False):
            return self.m.result_array()         # (exact mode: every rank settled the same record, host backend included)
        return self._folded

    def result(self):
        out = self.result_array()
        return None if out is None else tuple(out)
