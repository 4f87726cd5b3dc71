"""Template matching search drivers (host side) over the HIP engine.

Drop-in for the reference's ``scarplet/core.py`` API on the hot path:

    match(data, Template, **kwargs)                      core.py:266-294
    calculate_best_fit_parameters(dem, Template, ...)    core.py:139-195
    calculate_best_fit_parameters_serial(...)            core.py:65-136
    match_template(data, Template, scale, age, angle)    core.py:297-377
    compare(results, ny, nx)                             core.py:198-243
    load(filename)                                       core.py:246-263

Same names, arguments and return shapes.  What changed is where the work
happens: the reference fans ``match_template`` out over a process pool, one
full-size FFT convolution per (age, orientation), and folds the results on
the host; here a search is ONE call into libscarplet_hip.so - every template
becomes a small descriptor, the GPU synthesises the templates, correlates
them against the curvature and keeps the per-cell best (amp, age, angle, snr)
on the device.

Differences a caller can observe (DESIGN.md "Parity"):
  * amp / snr are computed in float32 on the device and returned as float64;
  * templates are folded orientation-major (the reference's ``match`` folds
    age-major); the two orders differ only where two templates tie exactly,
    which in float arithmetic is rounding noise in the reference as well;
  * a DEM with NaNs gives the reference's all-NaN maps (NaN spreads to every
    cell through its whole-grid FFTs) with a warning and without a device
    pass: call ``data._fill_nodata()`` first, as ``load`` does.
"""

import numpy as np

from scarplet_amd import _lib, _plan
from scarplet_amd import WindowedTemplate as _WT
from scarplet_amd.dem import DEMGrid

__all__ = ["match", "match_scales", "match_template", "compare", "load",
           "calculate_best_fit_parameters",
           "calculate_best_fit_parameters_serial", "Matcher"]

_CONTEXTS = {}


def _context(device):
    ctx = _CONTEXTS.get(device)
    if ctx is None:
        ctx = _lib.Context(device)
        _CONTEXTS[device] = ctx
    return ctx


def _grid_of(data):
    z = np.asarray(data._griddata)
    if z.ndim != 2:
        raise ValueError("data._griddata must be a 2-D array")
    gi = data._georef_info
    dx = float(gi.dx)
    dy = float(gi.dy if gi.dy is not None else gi.dx)
    return z, dx, dy


class Matcher(object):
    """A DEM resident on one GPU plus the running-best record of a search.

    ``match`` & co. build one per call; benchmarks and multi-search workflows
    keep one alive so the DEM stays in HBM."""

    def __init__(self, data=None, device=0, ctx=None):
        self.ctx = ctx if ctx is not None else _context(device)
        self.templates = None
        if data is not None:
            self.set_data(data)

    # -- DEM ------------------------------------------------------------------
    def set_data(self, data):
        z, dx, dy = _grid_of(data)
        # The block goes to the device as it is (56 GB/s from pageable memory here: 14 ms for
        # 10000 x 10000) and the device looks at it there (sc_dem_info): how many cells are NaN,
        # and whether it is the very block the context already holds - then the curvature planes
        # and the curvature spectra of the last search stay (one sl.match per scale on the same
        # data).  Two host passes over the DEM (np.isnan, a hash) used to cost 53 ms of every call.
        self.ny, self.nx = z.shape
        self.de = dx
        self.dx, self.dy = dx, dy
        self.core = (0, self.ny, 0, self.nx)
        self.whole = True
        self._z = z                              # (a reference: exact=True cuts the blocks it re-scores out of it)
        self.ctx.set_dem(z, dx, dy, _WT.centred_axis(self.nx, dx),
                         _WT.centred_axis(self.ny, dx))
        # A NaN anywhere in the DEM turns every reference output into NaN: the
        # curvature keeps the NaN (dem.py:85-86, 105) and the whole-grid FFTs
        # spread it to every cell (core.py:349-363).  Nothing is left to
        # compute, so such a DEM is never searched; the drivers answer
        # with the maps the reference returns (_nan_maps / _nan_fold).
        self.nan_dem = self.ctx.dem_nan > 0
        if self.nan_dem:
            import warnings
            warnings.warn("DEM contains NaN cells: every template's amplitude and SNR are "
                          "NaN (as in the reference); fill them first (DEMGrid._fill_nodata)")

    def set_block(self, z_dev_or_host, origin, shape, core, dx, dy,
                  block_shape=None):
        """Multi-GPU: a halo-extended block of a larger DEM (dist.py)."""
        self.ny, self.nx = shape
        self.de = dx
        self.dx, self.dy = dx, dy
        self.core = tuple(core)
        self.whole = False
        xa = _WT.centred_axis(self.nx, dx)
        ya = _WT.centred_axis(self.ny, dx)
        if isinstance(z_dev_or_host, np.ndarray):
            self.ctx.set_dem(z_dev_or_host, dx, dy, xa, ya, origin=origin,
                             shape=shape, core=core, wrap=False)
        else:
            ly, lx = block_shape
            self.ctx.set_dem_device(z_dev_or_host, ly, lx, dx, dy, xa, ya,
                                    origin, shape, core)

    # -- DEMs with NaN cells: the reference's (degenerate) outputs --------------
    def _nan_maps(self, t):
        """match_template() on a DEM with NaNs (core.py:348-375): amp and snr
        are NaN everywhere, then snr[err_mask] = 0, amp[lim] = snr[lim] = 0."""
        amp = np.full((self.ny, self.nx), np.nan)
        snr = np.full((self.ny, self.nx), np.nan)
        if hasattr(t, "get_err_mask"):
            snr[np.asarray(t.get_err_mask(), dtype=bool)] = 0
        lim = np.asarray(t.get_window_limits(), dtype=bool)
        amp[lim] = 0
        snr[lim] = 0
        return amp, snr

    def _nan_fold(self, Template, scale, params, angles, **kwargs):
        """compare() over such maps (core.py:228-240): 0*NaN poisons amp and
        snr of every cell some template leaves unmasked; age and angle stay 0
        (0 * finite); fully masked cells keep the zero record."""
        fast = self._nan_fold_builtin(Template, scale, params, angles) if not kwargs else None
        if fast is not None:
            return fast
        amp = np.zeros((self.ny, self.nx))
        snr = np.zeros((self.ny, self.nx))
        for ang in angles:
            for par in params:
                a, s = self._nan_maps(Template(scale, par, ang, self.nx, self.ny, self.de, **kwargs))
                amp[np.isnan(a)] = np.nan
                snr[np.isnan(s)] = np.nan
        zero = np.zeros((self.ny, self.nx))
        return np.stack([amp, zero, zero.copy(), snr])

    def _nan_fold_builtin(self, Template, scale, params, angles):
        """_nan_fold for the built-in classes without a template object (and two full-grid
        masks) per (age, orientation): a template leaves exactly its window-limit rectangle
        unmasked (grid_descriptors' ilo..ihi x jlo..jhi), so the cells that turn NaN are the
        union of those rectangles - one pass over a corner-count grid - and, for the UpperBreak
        classes, the part of each orientation's union outside its error half-plane
        (WT.py:257-267, 294-304).  None for any other class."""
        from scarplet_amd.WindowedTemplate import grid_descriptors, centred_axis, _trig, \
            FLAG_ERR_XR_LE0, FLAG_ERR_XR_GE0
        g = grid_descriptors(Template, scale, params, angles, self.nx, self.ny, self.de)
        if g is None:
            return None
        ny, nx = self.ny, self.nx

        def union(ilo, ihi, jlo, jhi):
            keep = (ihi >= ilo) & (jhi >= jlo)
            cnt = np.zeros((ny + 1, nx + 1), dtype=np.int32)
            for a, b, sgn in ((ilo, jlo, 1), (ilo, jhi + 1, -1), (ihi + 1, jlo, -1), (ihi + 1, jhi + 1, 1)):
                np.add.at(cnt, (a[keep], b[keep]), sgn)
            return cnt.cumsum(0).cumsum(1)[:ny, :nx] > 0

        ilo, ihi, jlo, jhi = (np.asarray(g[k]) for k in ("ilo", "ihi", "jlo", "jhi"))
        amp_nan = union(ilo.ravel(), ihi.ravel(), jlo.ravel(), jhi.ravel())
        flags = int(g["flags"])
        if flags & (FLAG_ERR_XR_LE0 | FLAG_ERR_XR_GE0):
            x = centred_axis(nx, self.de)[np.newaxis, :]
            y = centred_axis(ny, self.de)[:, np.newaxis]
            snr_nan = np.zeros((ny, nx), dtype=bool)
            for ib, ang in enumerate(angles):
                ca, sa = _trig(-ang)[:2]
                xr = x * ca + y * sa
                err = (xr <= 0) if flags & FLAG_ERR_XR_LE0 else (xr >= 0)
                snr_nan |= union(ilo[ib], ihi[ib], jlo[ib], jhi[ib]) & ~err
        else:
            snr_nan = amp_nan
        zero = np.zeros((ny, nx))
        return np.stack([np.where(amp_nan, np.nan, 0.0), zero, zero.copy(), np.where(snr_nan, np.nan, 0.0)])

    # -- templates --------------------------------------------------------------
    def describe(self, Template, scale, params, angles, id_base=0, id_of=None, **kwargs):
        """Descriptors for the (param, angle) grid, orientation-major.
        Template (param ia, angle ib) gets id ``id_base + ia * n_angles + ib``, or
        ``id_of(ia, ib)`` (dist.OrientationMatcher numbers them in fold order).
        Returns (ctypes array, support bbox union, largest tap count): the taps of a built-in
        window are estimated as the lattice points of its c x d rectangle (at most its box),
        a generic window's are counted."""
        n_par, n_ang = len(params), len(angles)
        arr = (_lib.sc_template * (n_par * n_ang))()
        fast = self._describe_grid(arr, Template, scale, params, angles, id_base, id_of) \
            if not kwargs and n_par * n_ang > 0 else None
        if fast is not None:
            return (arr,) + fast
        boxes = []
        k = 0
        max_area = 0
        import time
        t_start = time.perf_counter()
        for ib, ang in enumerate(angles):
            cc, sc2, ss = _plan.curvature_coefficients(ang)
            for ia, par in enumerate(params):
                t = Template(scale, par, ang, self.nx, self.ny, self.de,
                             **kwargs)
                desc = t._device_descriptor() \
                    if hasattr(t, "_device_descriptor") else None
                if desc is None:
                    # one full-grid template() and up to two full-grid masks at a time: t, W and
                    # the masks of the previous template are dropped before the next is built
                    desc = self._describe_generic(t)
                    del t
                    if k == 0:
                        self._warn_generic_cost(time.perf_counter() - t_start, n_par * n_ang)
                s = arr[k]
                s.kind, s.flags = desc["kind"], desc["flags"]
                s.cos_a, s.sin_a = desc["cos_a"], desc["sin_a"]
                s.c, s.d, s.p0, s.p1 = desc["c"], desc["d"], desc["p0"], desc["p1"]
                s.cc, s.sc2, s.ss = cc, sc2, ss
                s.ilo, s.ihi, s.jlo, s.jhi = desc["limits"]
                s.pmin, s.pmax, s.qmin, s.qmax = desc["bbox"]
                s.id = id_base + ia * n_ang + ib if id_of is None else int(id_of(ia, ib))
                s.window = desc.get("window", -1)
                if s.pmax < s.pmin or s.qmax < s.qmin:
                    # empty support: a 1-cell box of zeros keeps the kernels
                    # uniform (W == 0 everywhere -> amp = nan/0 like numpy)
                    s.pmin = s.pmax = s.qmin = s.qmax = 0
                boxes.append((s.pmin, s.pmax, s.qmin, s.qmax))
                box = (s.pmax - s.pmin + 1) * (s.qmax - s.qmin + 1)
                if s.kind == _WT.KIND_WINDOW:
                    taps = int(s.p0)
                else:
                    c_eff = min(float(s.c), np.sqrt(_WT.EXP_UNDERFLOW) / abs(s.p0)) \
                        if (s.kind == _WT.KIND_RICKER and s.p0 != 0) else float(s.c)
                    taps = int((2 * c_eff / self.de + 1) * (2 * float(s.d) / self.de + 1))
                max_area = max(max_area, min(box, taps))
                k += 1
        return arr, _plan.bbox_union(boxes), max_area

    def _describe_grid(self, arr, Template, scale, params, angles, id_base, id_of):
        """describe() for the built-in classes without a Python object per template
        (WindowedTemplate.grid_descriptors): fills ``arr`` through a numpy view of the struct
        array.  Returns (bbox union, max taps), or None when the class is not a built-in."""
        from scarplet_amd.WindowedTemplate import grid_descriptors
        g = grid_descriptors(Template, scale, params, angles, self.nx, self.ny, self.de)
        if g is None:
            return None
        n_par, n_ang = len(params), len(angles)
        T = _lib.sc_template
        names = [f[0] for f in T._fields_]
        dt = np.dtype({"names": names,
                       "formats": [np.dtype(f[1]) for f in T._fields_],
                       "offsets": [getattr(T, n).offset for n in names],
                       "itemsize": _lib.C.sizeof(T)})
        v = np.frombuffer(arr, dtype=dt).reshape(n_ang, n_par)
        v["kind"], v["flags"], v["window"] = g["kind"], g["flags"], -1
        for k in ("cos_a", "sin_a", "c", "d", "p0", "p1", "ilo", "ihi", "jlo", "jhi"):
            v[k] = g[k]
        cf = np.array([_plan.curvature_coefficients(a) for a in angles], dtype=np.float64)
        v["cc"], v["sc2"], v["ss"] = cf[:, 0:1], cf[:, 1:2], cf[:, 2:3]
        ia, ib = np.meshgrid(np.arange(n_par), np.arange(n_ang))
        v["id"] = id_base + ia * n_ang + ib if id_of is None else id_of(ia, ib)
        pmin, pmax, qmin, qmax = (np.array(g[k]) for k in ("pmin", "pmax", "qmin", "qmax"))
        empty = (pmax < pmin) | (qmax < qmin)           # see the loop below
        for a_ in (pmin, pmax, qmin, qmax):
            a_[empty] = 0
        v["pmin"], v["pmax"], v["qmin"], v["qmax"] = pmin, pmax, qmin, qmax
        bbox = (min(0, int(pmin.min())), max(0, int(pmax.max())),
                min(0, int(qmin.min())), max(0, int(qmax.max())))
        box = (pmax - pmin + 1) * (qmax - qmin + 1)
        c_eff = np.asarray(g["c"], dtype=np.float64)
        if int(g["kind"]) == _WT.KIND_RICKER:
            with np.errstate(divide="ignore"):
                c_eff = np.minimum(c_eff, np.sqrt(_WT.EXP_UNDERFLOW) / np.abs(np.asarray(g["p0"])))
        taps = (2 * c_eff / self.de + 1) * (2 * np.asarray(g["d"]) / self.de + 1)
        area = int(np.minimum(box, taps).max())
        return bbox, area

    # projected host seconds of a generic-plugin search above which describe() says so
    GENERIC_WARN_SECONDS = 60.0

    def _warn_generic_cost(self, first_seconds, n_templates):
        """A plugin that does not describe itself to the device is evaluated on the host: one
        full-grid numpy template() (+ masks) per (age, orientation) - 800 MB and seconds each at
        10000 x 10000.  Said once per search, from the measured cost of the first template."""
        total = first_seconds * n_templates
        if total > self.GENERIC_WARN_SECONDS:
            import warnings
            warnings.warn("generic template plugin: template() and the masks are evaluated on the host for "
                          "each of the %d templates (%.2f s for the first on this %d x %d grid: about %.0f s "
                          "in all, one full-grid array at a time); the built-in classes - this package's or "
                          "the reference's own - are synthesised on the device instead"
                          % (n_templates, first_seconds, self.ny, self.nx, total))

    def _describe_generic(self, t):
        """Any WindowedTemplate-like plugin: evaluate its numpy methods on the
        host and upload the non-zero window and its masks
        (plugin contract, core.py:345-346, 369-375)."""
        W = np.asarray(t.template(), dtype=float)
        if W.shape != (self.ny, self.nx):
            raise ValueError("template() must return an (ny, nx) array")
        nz = np.nonzero(W)
        if nz[0].size == 0:
            k0 = k1 = self.ny // 2
            l0 = l1 = self.nx // 2
        else:
            k0, k1 = int(nz[0].min()), int(nz[0].max())
            l0, l1 = int(nz[1].min()), int(nz[1].max())
        win = W[k0:k1 + 1, l0:l1 + 1]
        slot = self.ctx.upload_window(win)
        lim = np.asarray(t.get_window_limits(), dtype=bool)
        err = np.asarray(t.get_err_mask(), dtype=bool) \
            if hasattr(t, "get_err_mask") else None
        limits = (0, self.ny - 1, 0, self.nx - 1)
        lim_mask = None
        if lim.any():
            keep_r = np.nonzero(~lim.all(axis=1))[0]
            keep_c = np.nonzero(~lim.all(axis=0))[0]
            rect = np.ones_like(lim)
            if keep_r.size and keep_c.size:
                rect[keep_r[0]:keep_r[-1] + 1, keep_c[0]:keep_c[-1] + 1] = False
                limits = (int(keep_r[0]), int(keep_r[-1]),
                          int(keep_c[0]), int(keep_c[-1]))
            else:
                limits = (0, -1, 0, -1)
            if not np.array_equal(rect, lim):
                lim_mask = lim
        if lim_mask is not None or err is not None:
            self.ctx.set_masks(slot, lim_mask, err)
        alpha = getattr(t, "alpha", 0.0) or 0.0
        return dict(kind=_WT.KIND_WINDOW, flags=0, cos_a=float(np.cos(alpha)),
                    sin_a=float(np.sin(alpha)), c=0.0, d=0.0,
                    p0=float(np.count_nonzero(win)),
                    p1=float(np.sum(W ** 2)), limits=limits,
                    bbox=(k0 - self.ny // 2, k1 - self.ny // 2,
                          l0 - self.nx // 2, l1 - self.nx // 2), window=slot)

    # -- planning -----------------------------------------------------------------
    def plan_for(self, bbox, max_area, method="auto", group=None, n_params=1):
        if method == "auto":
            direct_ok = _plan.direct_window_fits(bbox[3] - bbox[2] + 1)
            try:
                fft = _plan.Plan(self.ny, self.nx, self.core, bbox,
                                 whole=self.whole, method=_plan.METHOD_FFT)
            except ValueError:
                if not direct_ok:         # larger than the largest tile AND the LDS slab
                    raise
                fft = None
            n_cells = (self.core[1] - self.core[0]) * (self.core[3] - self.core[2])
            method = "direct" if fft is None or (
                direct_ok and _plan.direct_cost(max_area) < _plan.fft_cost(fft, n_cells, n_params)) else "fft"
        m = _plan.METHOD_DIRECT if method == "direct" else _plan.METHOD_FFT
        p = _plan.Plan(self.ny, self.nx, self.core, bbox, whole=self.whole,
                       method=m)
        if group is None:
            # templates per inverse-transform launch: all parameters of one
            # orientation, capped so the intermediate planes stay under ~4 GB
            per_templ = 2 * 8 * max(p.Ty * p.Tx, 1)
            group = int(max(1, min(n_params, 64, 4e9 // per_templ)))
        p.group = group
        sp = _lib.sc_plan(method=m, Ty=p.Ty, Tx=p.Tx, Vy=p.Vy, Vx=p.Vx,
                          nty=p.nty, ntx=p.ntx, circ_y=int(p.circ_y),
                          circ_x=int(p.circ_x), Py=p.Py, Qx=p.Qx, group=group)
        return p, sp

    # -- searches -------------------------------------------------------------------
    # exact=True: relative window within which the FFT row pass flags near-ties (option "near_window"), by template
    # family: TWICE the largest float32 SNR error measured on the path, plus a tenth.  (With scores off by at most e, the
    # float64 argmax scores within 2 e of the final holder of the record, hence within 2 e of whatever held the record
    # when it was scored or displaced: it is named by an event or is the holder - DESIGN.md section 6.)  Scarp family:
    # e = 2.74e-4 (ONE search of round 6's 2 000-search fuzz on random-walk surfaces, profiles/r06_fuzz_oracle.txt; 1.6e-4
    # in round 5's, 4.3e-5 on the benchmark DEM - the window was 3.5e-4 until that search).  Ricker: e = 3.2e-4 (the int16
    # Grand Canyon DEM at scale 5: a Ricker window's support is the float64 underflow of its exponential - tiles with far
    # more energy, a larger float32 error; 4.9e-5 in the fuzz); round 5 flagged inside twice that (1.4e-3: a quarter of
    # that DEM's cells for one to decide).
    EXACT_WINDOW = {_WT.KIND_SCARP: 6e-4, _WT.KIND_RICKER: 7e-4}
    EXACT_MAX_COST = 50.0                        # re-scoring is not started beyond this many times the search's own cost
    EXACT_PATCH = (8, 256)                       # rows x columns re-scored around a flagged cell: one real-space workgroup

    def search(self, Template, scale, params, angles, method="auto",
               group=None, reset=True, sync=True, exact=False, **kwargs):
        """Fold every (param, angle) template into the running best.

        ``exact=True``: the argmax of EVERY cell is the float64 reference's.  The float32 paths are exact in (age,
        orientation) except where two templates score closer together than their arithmetic resolves - a handful of
        cells per million on a DEM with a noise floor (11 of 262 144 on the int16 Grand Canyon DEM, 502 of 1e8 on the
        benchmark DEM).  With the flag on, the search marks the cells where some template came within EXACT_WINDOW of the
        running best and lists which; the float64 argmax of a marked cell can only be among those templates, and exactly
        those (cell, template) pairs are scored in float64 on the device, which also picks the winner and writes it into
        the record (_search_exact, sc_settle_exact).  ``exact=None``: on for the built-in template classes, off for
        plugins whose windows the host uploads (no float64 form on the device).  Cost: 13 % on the row pass plus the
        pairs."""
        params = np.atleast_1d(np.asarray(params, dtype=float))
        angles = np.atleast_1d(np.asarray(angles, dtype=float))
        self._patches = []
        self._cells64 = None
        if getattr(self, "nan_dem", False):
            self._nan_result = self._nan_fold(Template, scale, params, angles, **kwargs)
            self.params, self.angles = params, angles
            return self
        self._nan_result = None
        # ids are cumulative over the searches folded into one record: a later
        # search with another grid must not re-use the ids cells already hold
        if reset or getattr(self, "_id_par", None) is None:
            self._id_par, self._id_ang = np.empty(0), np.empty(0)
        arr, bbox, max_area = self.describe(Template, scale, params, angles,
                                            id_base=len(self._id_par), **kwargs)
        self.plan, sp = self.plan_for(bbox, max_area, method, group,
                                      n_params=len(params))
        if reset:
            self.ctx.reset_best()
        # the host block of the (4, h, w) float64 result, faulted in by a thread of its own while the device searches
        # (a first call otherwise pays 75 ms per 3.2 GB for fresh pages under the device-to-host copy)
        pre = None
        if sync and reset:
            from scarplet_amd import _hostpool
            pre = _hostpool.prefault((4, self.core[1] - self.core[0], self.core[3] - self.core[2]))
        if exact is None:
            exact = bool(reset and sync and len(arr) and all(int(arr[k].kind) != _WT.KIND_WINDOW for k in (0, len(arr) - 1)))
        if exact:
            if not (reset and sync):
                raise ValueError("exact=True needs reset=True and sync=True")
            self.exact_stats = {"flagged_cells": 0, "patches": 0, "changed_cells": 0, "float64_cells": 0}
            self._search_exact(arr, sp, bbox, max_area, method, group, Template, scale, params, angles, kwargs)
            del pre
            self.params, self.angles = params, angles
            self.n_templates = len(arr)
            self._id_par = np.concatenate([self._id_par, np.repeat(params, len(angles))])
            self._id_ang = np.concatenate([self._id_ang, np.tile(angles, len(params))])
            return self
        self.ctx.match(arr, sp, sync=sync)
        # (not joined: a search shorter than the touching finds the block still referenced by the thread and takes a
        #  fresh one, as before - never slower than without)
        del pre
        self.method_used = "direct" if sp.method == _plan.METHOD_DIRECT else "fft"
        if method == "auto" and reset and sync and self.method_used == "fft":
            self._exact_path_if_unresolved(arr, bbox, max_area, group, len(params))
        elif method == "fft" and reset and sync:
            self._warn_if_unresolved()
        self.params, self.angles = params, angles
        self.n_templates = len(arr)
        self._id_par = np.concatenate([self._id_par, np.repeat(params, len(angles))])
        self._id_ang = np.concatenate([self._id_ang, np.tile(angles, len(params))])
        return self

    def run_described(self, arr, sp, exact_window=0.0, n_twin=0):
        """One search of descriptors already built (``describe`` + ``plan_for``): reset, sc_match and - with
        ``exact_window`` > 0 - the float64 settle of its near-ties (sc_settle_exact; ``n_twin``: templates of the last
        orientation that are the first orientation's, end_twins()).  What bench.py times as one step; returns the
        settle's counters or None."""
        self.ctx.reset_best()
        if not exact_window > 0.0:
            self.ctx.match(arr, sp, sync=True)
            return None
        self.ctx.set_option("near_window", float(exact_window))
        try:
            self.ctx.match(arr, sp, sync=True)
        finally:
            self.ctx.set_option("near_window", 0.0)
        return self.ctx.settle_exact(n_twin, self.EXACT_MAX_F64)

    def exact_window_for(self, arr, sp):
        """The near-tie window exact=True searches these descriptors with: by path and template family."""
        if sp.method != _plan.METHOD_FFT:
            return self.EXACT_WINDOW_DIRECT
        kinds = {int(arr[0].kind), int(arr[len(arr) - 1].kind)}
        return max(self.EXACT_WINDOW.get(k, max(self.EXACT_WINDOW.values())) for k in kinds)

    def end_twins(self, arr, n_params, angles):
        """How many templates at the end of the orientation-major list ``arr`` repeat its first ones: n_params where the
        grid runs from -pi/2 to +pi/2 and the class is one of the symmetric built-ins (_without_end_twin), else 0."""
        return len(arr) - len(self._without_end_twin(arr, n_params, angles))

    def _warn_if_unresolved(self):
        """The FFT path was asked for by name: it is handed out as it is, but not silently where the device's own
        statistic says it cannot resolve this surface in float32."""
        wins, near = self.ctx.resolution_stats()
        self.unresolved_frac = near / wins if wins else 0.0
        if self.unresolved_frac > self.UNRESOLVED_MAX:
            import warnings
            warnings.warn("method='fft': %.1f %% of the cells this search won lie within the float32 "
                          "resolution floor of the FFT convolution (a surface without a noise floor of its "
                          "own); their argmax is rounding noise - method='auto' or 'direct' gives the exact "
                          "real-space answer" % (100 * self.unresolved_frac))

    def _search_exact(self, arr, sp, bbox, max_area, method, group, Template, scale, params, angles, kwargs):
        """exact=True (round 6: settled on the device).  The search runs with option "near_window" on - either path then
        flags its near-ties and lists them as events (cell, template scored, holder of the record) - and sc_settle_exact
        scores exactly the (cell, template) pairs those events name in float64 and gives every flagged cell its float64
        argmax: the record's (age, orientation) becomes the reference's (compare(), core.py:230-240, folds float64 maps),
        the float64 (amp, snr) ride along as patches under the result.  No host pass over the planes.

        Which path: the FFT tiles where the planner chose them and the row kernel can flag (no per-cell masks); the
        real-space path otherwise - by name, by `auto` for a small support, for UpperBreak's error masks, or because the
        device's own statistic says the FFT convolution cannot resolve this surface in float32 (method="auto", as without
        the mode).  A list that overflows, or more float64 work than EXACT_MAX_F64, takes the longer routes of round 5."""
        import warnings
        n_par = len(params)
        n_twin = self.end_twins(arr, n_par, angles)
        fft = sp.method == _plan.METHOD_FFT
        win_fft = self.exact_window_for(arr, sp) if fft else 0.0
        try:
            if fft:
                self.ctx.set_option("near_window", win_fft)
                try:
                    self.ctx.match(arr, sp, sync=True)
                except _lib.ScarpletHipError as e:
                    # per-cell masks (generic plugins, UpperBreak error masks) or a tile size the flagging row kernel is
                    # not built for: the whole search on the real-space path instead
                    if "near-tie flags" not in str(e):
                        raise
                    fft = False
                if fft and method == "auto":
                    wins, near = self.ctx.resolution_stats()
                    self.unresolved_frac = near / wins if wins else 0.0
                    if self.unresolved_frac > self.UNRESOLVED_MAX:
                        note = ("the FFT path cannot resolve %.1f %% of this surface's cells in float32 (no noise floor "
                                "of its own)" % (100 * self.unresolved_frac))
                        ww = bbox[3] - bbox[2] + 1
                        n_cells = (self.core[1] - self.core[0]) * (self.core[3] - self.core[2])
                        if not _plan.direct_window_fits(ww) or \
                                _plan.direct_cost(max_area) > 100 * _plan.fft_cost(self.plan, n_cells, n_par):
                            warnings.warn(note + "; method='direct' is exact but much slower here - not taken automatically")
                        else:
                            warnings.warn(note + ": searched again on the exact real-space path")
                            fft = False
                elif fft and method == "fft":
                    self._warn_if_unresolved()
                if not fft:
                    self.plan, sp = self.plan_for(bbox, max_area, "direct", group, n_params=n_par)
                    self.ctx.reset_best()
            if not fft:
                self.ctx.set_option("near_window", self.EXACT_WINDOW_DIRECT)
                self.ctx.match(arr, sp, sync=True)
        finally:
            self.ctx.set_option("near_window", 0.0)
        self.method_used = "fft" if fft else "direct"
        if self.EXACT_USE_EVENTS:
            try:
                st = self.ctx.settle_exact(n_twin, self.EXACT_MAX_F64)
                self.exact_stats.update(st, route="device")
                return
            except _lib.ScarpletHipError as e:
                msg = str(e)
                if "built-in templates only" in msg:
                    warnings.warn("exact=True: a plugin's window exists in float32 on the device only - the float32 result "
                                  "stands (the built-in template classes are settled in float64)")
                    self.exact_stats["skipped"] = True
                    return
                if "overflowed" not in msg and "too much float64 work" not in msg:
                    raise
                self.exact_stats["settle"] = msg
        # the longer routes (round 5): host lists
        if not self.whole:
            warnings.warn("exact=True: the near-ties of this block were not settled (%s)" % self.exact_stats.get("settle", "host route off"))
            self.exact_stats["skipped"] = True
            return
        if fft:
            self._rescore_near_ties(Template, scale, params, angles, kwargs)
        else:
            last = [tuple(c) for c in np.argwhere(self.ctx.near_ties())]
            self.exact_stats["flagged_cells"] = len(last)
            self._score_float64([(i + self.core[0], j + self.core[2]) for i, j in last], arr, bbox)

    @staticmethod
    def _without_end_twin(arr, n_params, angles):
        """The descriptors (orientation-major) without the LAST orientation where that is +pi/2 and the first is -pi/2
        and the templates are the symmetric built-ins (Scarp: W(alpha + pi) = -W(alpha); Ricker / Channel: the same
        template): the two ends of the reference's grid (core.py:173-175) are one template, their float64 SNRs differ
        by rounding noise (1e-13: one maximum by the parity policy) and their float32 real-space scores are the same
        bits - which the real-space path's own flags (equal scores included) would hand to float64 cell by cell for
        nothing.  The real-space steps of exact=True search one of the two."""
        from scarplet_amd import WindowedTemplate as _WT
        n = len(arr) - n_params
        if len(angles) < 2 or n <= 0 or abs(angles[0] + np.pi / 2) > 1e-12 or abs(angles[-1] - np.pi / 2) > 1e-12:
            return arr
        err = _WT.FLAG_ERR_XR_LE0 | _WT.FLAG_ERR_XR_GE0
        if any(int(arr[k].window) >= 0 or (int(arr[k].flags) & err) for k in (0, len(arr) - 1)):
            return arr
        out = (_lib.sc_template * n)()
        for k in range(n):
            out[k] = arr[k]
        return out

    def _direct_exact(self, arr, sp):
        """exact=True on the real-space path: the search with its own near-tie flags on (EXACT_WINDOW_DIRECT: the cells
        it decides inside its float32 rounding); returns those cells for _score_float64."""
        self.ctx.set_option("near_window", self.EXACT_WINDOW_DIRECT)
        try:
            self.ctx.reset_best()
            self.ctx.match(arr, sp, sync=True)
            return [tuple(c) for c in np.argwhere(self.ctx.near_ties())]
        finally:
            self.ctx.set_option("near_window", 0.0)

    def _rescore_near_ties(self, Template, scale, params, angles, kwargs):
        """exact=True, second and third step: the cells the FFT row pass flagged are searched again on the
        real-space path - which flags, in turn, the cells it decides inside ITS float32 rounding
        (EXACT_WINDOW_DIRECT) - and those last cells are scored in float64 (sc_score_cells_f64: every template,
        the reference's own arithmetic) and take the float64 argmax."""
        from scarplet_amd import dist as _dist
        flags = self.ctx.near_ties()
        cells = np.argwhere(flags)
        self.exact_stats = {"flagged_cells": int(len(cells)), "patches": 0, "changed_cells": 0, "float64_cells": 0}
        self._cells64 = None
        if not len(cells):
            return
        ph, pw = self.EXACT_PATCH
        todo = sorted({(int(i) // ph, int(j) // pw) for i, j in cells})
        # what the second pass would cost against the search it follows (the planner's per-cell figures): a large DEM
        # with thousands of flagged cells and templates of tens of thousands of taps is hours of real-space work -
        # said, not started
        n_t = len(params) * len(angles)
        _, bbox0, area0 = self.describe(Template, scale, params[:1], angles[:1], **kwargs)
        area = max(area0, int((2 * float(np.max(np.abs(self.plan.bbox))) + 1) ** 2 * 0.15))
        n_cells = self.ny * self.nx
        cost_fft = _plan.fft_cost(self.plan, n_cells, len(params)) * n_cells * n_t
        cost_patches = _plan.direct_cost(area) * min(len(todo) * ph * pw, n_cells) * n_t
        if cost_patches > self.EXACT_MAX_COST * cost_fft:
            import warnings
            warnings.warn("exact=True: %d cells flagged in %d patches; searching them again on the real-space path would "
                          "take about %.0f times the search itself - not done, the FFT result stands (Matcher.ctx.near_ties() "
                          "has the flags)" % (len(cells), len(todo), cost_patches / cost_fft))
            self.exact_stats["skipped"] = True
            return
        arr_main, bbox, max_area = self.describe(Template, scale, params, angles, **kwargs)
        last = []                                        # global (i, j) of the cells the real-space path leaves undecided
        if len(todo) * ph * pw > 0.5 * self.ny * self.nx:
            # the flags cover half the DEM: a surface the FFT path does not resolve (or a template family whose SNR
            # varies slowly with the orientation: a Ricker) - what method="auto" answers with a whole real-space
            # search; per-patch searches would cost more than that
            import warnings
            warnings.warn("exact=True: %d cells flagged (%d patches): searching the whole DEM on the real-space "
                          "path instead" % (len(cells), len(todo)))
            self.plan, sp = self.plan_for(bbox, max_area, "direct", None, n_params=len(params))
            arr_main = self._without_end_twin(arr_main, len(params), angles)
            last = self._direct_exact(arr_main, sp)
            self.method_used = "direct"
        else:
            aux = getattr(self, "_aux", None)
            if aux is None:
                aux = self._aux = Matcher(ctx=_lib.Context(self.ctx.device))
            z = np.asarray(self._z)
            ny, nx = self.ny, self.nx
            aux.ny, aux.nx, aux.de = ny, nx, self.de
            arr, _, _ = aux.describe(Template, scale, params, angles, **kwargs)
            arr = self._without_end_twin(arr, len(params), angles)
            halo = _dist.halo_for_search(bbox, ny, nx)
            aux.ctx.set_option("near_window", self.EXACT_WINDOW_DIRECT)
            try:
                for (bi, bj) in todo:
                    i0, j0 = bi * ph, bj * pw
                    i1, j1 = min(i0 + ph, ny), min(j0 + pw, nx)
                    gi = np.arange(i0 - halo[0], i1 + halo[1]) % ny      # the DEM is a torus (the reference's circular convolution)
                    gj = np.arange(j0 - halo[2], j1 + halo[3]) % nx
                    blk = np.ascontiguousarray(z[np.ix_(gi, gj)], dtype=np.float64)
                    aux.set_block(blk, (i0 - halo[0], j0 - halo[2]), (ny, nx), (i0, i1, j0, j1), self.dx, self.dy)
                    _, sp = aux.plan_for(bbox, max_area, "direct", None, n_params=len(params))
                    aux.ctx.reset_best()
                    aux.ctx.match(arr, sp, sync=True)
                    amp, snr, idx = aux.ctx.get_best()
                    sel = flags[i0:i1, j0:j1] != 0
                    self._patches.append((i0, j0, sel, amp, snr, idx))
                    self.exact_stats["patches"] += 1
                    last += [(i0 + int(a), j0 + int(b)) for a, b in np.argwhere(sel & (aux.ctx.near_ties() != 0))]
            finally:
                aux.ctx.set_option("near_window", 0.0)
                aux.ctx.clear_windows()
        self._score_float64(last, arr_main, bbox)

    EXACT_USE_EVENTS = True                      # (False: always the longer, host-side routes - tests, comparisons)

    # exact=True on the real-space path: the window of its near-tie flags and events - twice its largest measured SNR error
    # plus a tenth (2.98e-4 in ONE search of round 6's 4 100-search fuzz - a single old-scarp template, thousands of taps;
    # 2.15e-4 in another; 1.0e-4 in round 5's, 4e-5 on the tests' DEMs) - and how much float64 work is started at most
    EXACT_WINDOW_DIRECT = 6.6e-4
    EXACT_MAX_F64 = 2e12                         # (cell, template) pairs x support-box cells (~3e11 a second; the C3 search: 4.8e11)

    def _score_float64(self, last, arr_main, bbox):
        """The cells the real-space path decided inside its own rounding: match_template() in float64 for every
        template (the MAIN context holds the whole DEM and the search's descriptors and sums), the argmax in fold
        order - ties to the earlier template, as the device folds."""
        if not last:
            return
        n_t = len(arr_main)
        box = (bbox[1] - bbox[0] + 1) * (bbox[3] - bbox[2] + 1)
        if float(len(last)) * n_t * box > self.EXACT_MAX_F64:
            import warnings
            warnings.warn("exact=True: %d cells are left to float64 by the real-space path; scoring them for %d templates is "
                          "not started (%.1e support cells) - they keep the real-space answer" % (len(last), n_t, float(len(last)) * n_t * box))
            return
        if self.method_used != "direct":
            # the main context's last search must be THIS search (descriptors, template sums): it is - the patches ran
            # in the auxiliary context
            pass
        cells = np.asarray(sorted(set(last)), dtype=np.int32).reshape(-1, 2)
        try:
            amp, snr = self.ctx.score_cells_f64(cells, n_t)
        except _lib.ScarpletHipError as e:
            if "built-in templates only" in str(e):
                return
            raise
        k = np.argmax(snr, axis=1)                       # (first maximum: the fold order is the order of arr_main)
        rows = np.arange(len(cells))
        ids = np.array([arr_main[int(v)].id for v in k], dtype=np.int64)
        best_snr, best_amp = snr[rows, k], amp[rows, k]
        self._cells64 = (cells[:, 0].astype(np.int64), cells[:, 1].astype(np.int64), best_amp, best_snr, ids)
        self.exact_stats["float64_cells"] = int(len(cells))

    def _apply_patches(self, out):
        """The re-scored cells of exact=True into a (4, h, w) result."""
        par, ang = self._id_par, self._id_ang
        for (i0, j0, sel, amp, snr, idx) in getattr(self, "_patches", ()):
            won = sel & (idx < len(par))
            safe = np.where(won, idx, 0)
            h, w = sel.shape
            view = out[:, i0:i0 + h, j0:j0 + w]
            changed = won & ((view[1] != par[safe]) | (view[2] != ang[safe]))
            self.exact_stats["changed_cells"] += int(changed.sum())
            view[0][won] = amp[won]
            view[1][won] = par[safe][won]
            view[2][won] = ang[safe][won]
            view[3][won] = snr[won]
        c64 = getattr(self, "_cells64", None)
        if c64 is not None:
            ii, jj, amp, snr, ids = c64
            won = snr > 0
            changed = won & ((out[1][ii, jj] != par[ids]) | (out[2][ii, jj] != ang[ids]))
            self.exact_stats["changed_cells"] += int(changed.sum())
            out[0][ii[won], jj[won]] = amp[won]
            out[1][ii[won], jj[won]] = par[ids[won]]
            out[2][ii[won], jj[won]] = ang[ids[won]]
            out[3][ii[won], jj[won]] = snr[won]
        return out

    # share of the cells an FFT search won whose residual lies near the transforms' float32
    # resolution floor (sc_get_resolution_stats) above which method="auto" takes the exact path
    UNRESOLVED_MAX = 0.01

    def _exact_path_if_unresolved(self, arr, bbox, max_area, group, n_params):
        """method="auto" on a surface WITHOUT a noise floor (synthetic scarps stored as float32:
        quantisation noise only away from the feature): a float32 FFT convolution resolves an
        output only to a fraction of its tile's energy, and where the residual T3 - T1 the SNR
        divides by sinks to that resolution the argmax over templates is rounding noise (8 % of
        the cells of such a surface, tests/test_gpu_parity.py).  The device counts those cells
        while it folds; when more than UNRESOLVED_MAX of the wins are such, the search is run
        again on the real-space path, which sums locally and has no such limit - unless that
        would take beyond a hundred times longer, then a warning says so."""
        wins, near = self.ctx.resolution_stats()
        self.unresolved_frac = near / wins if wins else 0.0
        if self.unresolved_frac <= self.UNRESOLVED_MAX:
            return
        import warnings
        note = ("the FFT path cannot resolve %.1f %% of this surface's cells in float32 (no noise floor "
                "of its own)" % (100 * self.unresolved_frac))
        ww = bbox[3] - bbox[2] + 1
        n_cells = (self.core[1] - self.core[0]) * (self.core[3] - self.core[2])
        if not _plan.direct_window_fits(ww) or \
                _plan.direct_cost(max_area) > 100 * _plan.fft_cost(self.plan, n_cells, n_params):
            warnings.warn(note + "; method='direct' is exact but much slower here - not taken automatically")
            return
        warnings.warn(note + ": searched again on the exact real-space path")
        self.plan, sp = self.plan_for(bbox, max_area, "direct", group, n_params=n_params)
        self.ctx.reset_best()
        self.ctx.match(arr, sp, sync=True)
        self.method_used = "direct"

    def result(self):
        """(amp, age, angle, snr) float64 maps of the core region."""
        # template id = i_param * n_angles + i_angle (describe()); the conversion
        # of the float32 record into the reference's float64 planes runs on the
        # device, the host receives the finished (4, h, w) array
        out = self.result_array()
        return (out[0], out[1], out[2], out[3])

    def result_array(self):
        """The same as one (4, h, w) float64 array."""
        if getattr(self, "_nan_result", None) is not None:
            return self._nan_result
        if getattr(self, "_id_par", None) is None or not len(self._id_par):
            # descriptors were sent by hand (ctx.match): one grid, ids from 0
            return self.ctx.get_result(np.repeat(self.params, len(self.angles)),
                                       np.tile(self.angles, len(self.params)))
        out = self.ctx.get_result(self._id_par, self._id_ang)
        if getattr(self, "_patches", None) or getattr(self, "_cells64", None) is not None:
            if hasattr(self, "exact_stats"):
                self.exact_stats["changed_cells"] = 0
            out = self._apply_patches(out)
        return out

    def search_scales(self, Template, scales, params, angles, method="auto", exact=None, **kwargs):
        """A multi-scale job (BASELINE config C5: Channel at five scales x 181 orientations; the reference runs it as one
        sl.match per scale on the same data, docs/source/examples/channels.ipynb - its 4-plane result has no scale
        plane): one search per scale on THIS matcher, whose context keeps every orientation's curvature spectra from
        the first scale on (option "spectra_mb": the later scales skip the curvature passes where the tile plan stays
        the same - the same bits either way).  Returns a list of (4, h, w) float64 arrays, one per scale."""
        out = []
        for sc in scales:
            self.search(Template, sc, params, angles, method=method, exact=exact, **kwargs)
            out.append(np.array(self.result_array()))      # (a copy: large results are views of a recycled host block)
        return out

    def match_template(self, Template, scale, age, angle, method="auto",
                       **kwargs):
        if getattr(self, "nan_dem", False):
            return self._nan_maps(Template(scale, age, angle, self.nx, self.ny, self.de, **kwargs))
        arr, bbox, max_area = self.describe(Template, scale, [age], [angle],
                                            **kwargs)
        self.plan, sp = self.plan_for(bbox, max_area, method)
        amp, snr = self.ctx.match_template(arr[0], sp)
        return amp.astype(np.float64), snr.astype(np.float64)


# ------------------------------------------------------------------------------
# reference API
# ------------------------------------------------------------------------------
def load(filename):
    """Load a DEM and fill its nodata cells (core.py:246-263)."""
    data = DEMGrid(filename)
    data._fill_nodata()
    return data


def match_template(data, Template, scale, age, angle, **kwargs):
    """Amplitude and SNR of one template at every cell (core.py:297-377).

    Returns ``(amp, age, angle, snr)`` with ``age`` and ``angle`` the scalar
    inputs, like the reference."""
    opts = {k: kwargs.pop(k) for k in ("device", "method") if k in kwargs}
    m = Matcher(data, device=opts.get("device", 0))
    try:
        amp, snr = m.match_template(Template, scale, age, angle,
                                    method=opts.get("method", "auto"), **kwargs)
    finally:
        m.ctx.clear_windows()
    return amp, age, angle, snr


def calculate_best_fit_parameters(dem, Template, scale, age,
                                  ang_max=np.pi / 2, ang_min=-np.pi / 2,
                                  **kwargs):
    """Best-fitting amplitude / orientation / SNR for one age over the
    one-degree orientation grid (core.py:139-195).  Returns a (4, ny, nx)
    array: amp, age, angle, snr.  Like the reference, extra keyword arguments
    are accepted but not forwarded to the template (core.py:145, 182)."""
    device = kwargs.pop("device", 0)
    method = kwargs.pop("method", "auto")
    exact = kwargs.pop("exact", None)
    m = Matcher(dem, device=device)
    try:
        m.search(Template, scale, [age], _plan.angle_grid(ang_min, ang_max),
                 method=method, exact=exact)
        return m.result_array()
    finally:
        m.ctx.clear_windows()


def calculate_best_fit_parameters_serial(dem, Template, scale,
                                         ang_max=np.pi / 2,
                                         ang_min=-np.pi / 2, **kwargs):
    """Full (age, orientation) search, orientation-major, forwarding extra
    keyword arguments to the template (core.py:65-136).  Returns the 4-tuple
    (best_amp, best_age, best_angle, best_snr)."""
    device = kwargs.pop("device", 0)
    method = kwargs.pop("method", "auto")
    exact = kwargs.pop("exact", None)
    m = Matcher(dem, device=device)
    try:
        m.search(Template, scale, _plan.age_grid(),
                 _plan.angle_grid(ang_min, ang_max), method=method, exact=exact, **kwargs)
        return m.result()
    finally:
        m.ctx.clear_windows()


def _reference_fold_one_age(m, Template, scale, age, ang_min, ang_max, method):
    """calculate_best_fit_parameters' fold as the reference writes it (core.py:180-195): every
    orientation's match_template maps through compare() - two strict compares, a tie zeroes the
    record (core.py:230-240) - in float64 on the device (sc_compare_*)."""
    from scarplet_amd import _fold
    ny, nx = m.ny, m.nx

    def maps():
        for angle in _plan.angle_grid(ang_min, ang_max):
            amp, snr = m.match_template(Template, scale, age, angle, method=method)
            yield amp, age, angle, snr
    return np.stack(_fold.compare(maps(), ny, nx, device=m.ctx.device))


def match(data, Template, **kwargs):
    """Match a template family to a DEM (core.py:266-294).

    With ``age=``: one age, returns a (4, ny, nx) array.  Without: the 35-age
    grid 10**arange(0, 3.5, 0.1), returns the 4-tuple (amp, age, angle, snr).
    Keyword arguments: ``scale``, ``age``, ``ang_max``, ``ang_min`` as in the
    reference, plus ``device=`` (GPU ordinal), ``method=`` ('auto', 'fft',
    'direct'), ``ages=`` (override the age grid), ``exact=`` (default: on for
    the built-in template classes - every cell's (age, orientation) is the
    float64 reference's argmax: the near-ties of the float32 search are scored
    in float64 on the device, Matcher.search; ``exact=False``: the float32
    search as it is, 5 - 10 % faster) and ``fold=``:

    ``fold="fused"`` (default): ONE device search, the running best folded in
    the kernels - ties keep the incumbent, orientation-major order.
    ``fold="reference"``: the reference's literal fold - one match_template per
    (age, orientation), orientations folded per age by compare() and the ages'
    results folded by compare() again (core.py:180-195, 288-292), a tie zeroing
    the record (core.py:230-240) - on float32 maps, so where the float64
    reference saw no tie this may see one.  One device pass and two 8-byte-per-
    cell transfers per template: for small DEMs and for inspecting ties."""
    fold = kwargs.pop("fold", "fused")
    if fold not in ("fused", "reference"):
        raise ValueError("fold must be 'fused' or 'reference'")
    if fold == "reference":
        device = kwargs.pop("device", 0)
        method = kwargs.pop("method", "auto")
        ages = kwargs.pop("ages", None)
        scale = kwargs.pop("scale")
        ang_max = kwargs.pop("ang_max", np.pi / 2)
        ang_min = kwargs.pop("ang_min", -np.pi / 2)
        m = Matcher(data, device=device)
        try:
            if 'age' in kwargs:
                return _reference_fold_one_age(m, Template, scale, kwargs['age'], ang_min, ang_max, method)
            from scarplet_amd import _fold
            # (a list, not a generator: the context holds ONE compare session at a time)
            per_age = [_reference_fold_one_age(m, Template, scale, age, ang_min, ang_max, method)
                       for age in (_plan.age_grid() if ages is None else ages)]
            return _fold.compare(per_age, m.ny, m.nx, device=m.ctx.device)
        finally:
            m.ctx.clear_windows()
    if 'age' in kwargs:
        return calculate_best_fit_parameters(data, Template, **kwargs)
    device = kwargs.pop("device", 0)
    method = kwargs.pop("method", "auto")
    exact = kwargs.pop("exact", None)
    ages = kwargs.pop("ages", None)
    scale = kwargs.pop("scale")
    ang_max = kwargs.pop("ang_max", np.pi / 2)
    ang_min = kwargs.pop("ang_min", -np.pi / 2)
    m = Matcher(data, device=device)
    try:
        m.search(Template, scale, _plan.age_grid() if ages is None else ages,
                 _plan.angle_grid(ang_min, ang_max), method=method, exact=exact)
        return m.result()
    finally:
        m.ctx.clear_windows()


def match_scales(data, Template, scales, **kwargs):
    """``match`` for several scales of one template family on one DEM (the reference: one sl.match call per scale,
    channels.ipynb): the DEM goes to the device once, the orientations' curvature spectra are computed once.  Keyword
    arguments as ``match`` (``age=`` or the 35-age grid, ``ang_min`` / ``ang_max``, ``method``, ``exact``, ``device``).
    Returns a list with one (4, ny, nx) array per scale - the planes of ``match``: amp, age, angle, snr."""
    device = kwargs.pop("device", 0)
    method = kwargs.pop("method", "auto")
    exact = kwargs.pop("exact", None)
    ages = kwargs.pop("ages", None)
    ang_max = kwargs.pop("ang_max", np.pi / 2)
    ang_min = kwargs.pop("ang_min", -np.pi / 2)
    params = [kwargs.pop("age")] if "age" in kwargs else (_plan.age_grid() if ages is None else ages)
    if kwargs:
        raise TypeError("match_scales: unexpected keyword arguments %s" % sorted(kwargs))
    m = Matcher(data, device=device)
    try:
        return m.search_scales(Template, scales, params, _plan.angle_grid(ang_min, ang_max), method=method, exact=exact)
    finally:
        m.ctx.clear_windows()


def compare(results, ny, nx):
    """Fold an iterable of ``(amp, age, angle, snr)`` results into the
    per-cell best by SNR (core.py:198-243): strict compares, a tie zeroes the
    record, NaN is sticky.  Runs on the GPU (float64, like the reference)."""
    from scarplet_amd import _fold
    return _fold.compare(results, ny, nx)
