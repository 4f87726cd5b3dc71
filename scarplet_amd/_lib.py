"""ctypes binding of libscarplet_hip.so (include/scarplet_hip.h).

The library is the only compute path: if it cannot be loaded, or no GPU is
visible, the matcher raises - there is no CPU fallback in this package.
"""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCARPLET_HIP_LIB is the ONE environment variable this package reads: a developer hook of the
# tools/ scripts to load another build of the same library (the -DSC_ABLATE timing build,
# tools/ablate.sh).  The engine itself - libscarplet_hip.so - reads no environment at all; its
# options go through sc_set_option.
LIB_PATH = os.environ.get("SCARPLET_HIP_LIB") or os.path.join(_HERE, "libscarplet_hip.so")

SC_OK = 0
ABI_VERSION = 9
ID_NONE = 0xFFFFFFFF
COMM_ID_BYTES = 128

K_NAMES = ("k_curv", "k_windows", "k_direct", "k_fwd_rows", "k_fwd_cols",
           "k_inv_cols", "k_inv_rows", "k_settle")
K_CURV, K_WINDOWS, K_DIRECT, K_FWD_ROWS, K_FWD_COLS, K_INV_COLS, K_INV_ROWS, K_SETTLE = range(8)

XFER_RECV, XFER_SEND, XFER_LOCAL = 0, 1, 2


class ScarpletHipError(RuntimeError):
    pass


class sc_template(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flags", C.c_int32),
                ("cos_a", C.c_double), ("sin_a", C.c_double),
                ("c", C.c_double), ("d", C.c_double),
                ("p0", C.c_double), ("p1", C.c_double),
                ("cc", C.c_double), ("sc2", C.c_double), ("ss", C.c_double),
                ("ilo", C.c_int32), ("ihi", C.c_int32),
                ("jlo", C.c_int32), ("jhi", C.c_int32),
                ("pmin", C.c_int32), ("pmax", C.c_int32),
                ("qmin", C.c_int32), ("qmax", C.c_int32),
                ("id", C.c_uint32), ("window", C.c_int32)]


class sc_plan(C.Structure):
    _fields_ = [("method", C.c_int32), ("Ty", C.c_int32), ("Tx", C.c_int32),
                ("Vy", C.c_int32), ("Vx", C.c_int32), ("nty", C.c_int32),
                ("ntx", C.c_int32), ("circ_y", C.c_int32),
                ("circ_x", C.c_int32), ("Py", C.c_int32), ("Qx", C.c_int32),
                ("group", C.c_int32)]


class sc_xfer(C.Structure):
    _fields_ = [("peer", C.c_int32), ("kind", C.c_int32),
                ("sy0", C.c_int32), ("sx0", C.c_int32),
                ("dy0", C.c_int32), ("dx0", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32)]


_P = C.c_void_p
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)

# every symbol include/scarplet_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "sc_abi_version": (C.c_int, []),
    "sc_build_id": (C.c_char_p, []),
    "sc_device_count": (C.c_int, []),
    "sc_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "sc_destroy": (None, [_P]),
    "sc_last_error": (C.c_char_p, [_P]),
    "sc_set_dem": (C.c_int, [_P, _dp] + [C.c_int] * 10 + [C.c_double] * 2
                   + [C.c_int, _dp, _dp]),
    "sc_dem_info": (C.c_int, [_P, C.POINTER(C.c_longlong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]),
    "sc_set_dem_device": (C.c_int, [_P, _P] + [C.c_int] * 10
                          + [C.c_double] * 2 + [C.c_int, _dp, _dp]),
    "sc_upload_window": (C.c_int, [_P, _dp, C.c_int, C.c_int,
                                   C.POINTER(C.c_int)]),
    "sc_set_masks": (C.c_int, [_P, C.c_int, _bp, _bp]),
    "sc_clear_windows": (C.c_int, [_P]),
    "sc_reset_best": (C.c_int, [_P]),
    "sc_match": (C.c_int, [_P, C.POINTER(sc_template), C.c_int,
                           C.POINTER(sc_plan)]),
    "sc_match_async": (C.c_int, [_P, C.POINTER(sc_template), C.c_int,
                                 C.POINTER(sc_plan)]),
    "sc_sync": (C.c_int, [_P]),
    "sc_match_template": (C.c_int, [_P, C.POINTER(sc_template),
                                    C.POINTER(sc_plan), _fp, _fp]),
    "sc_get_best": (C.c_int, [_P, _fp, _fp, _up]),
    "sc_get_result": (C.c_int, [_P, _dp, _dp, C.c_int, _dp]),
    "sc_compare_begin": (C.c_int, [_P, C.c_int, C.c_int]),
    "sc_compare_fold": (C.c_int, [_P, _dp, _dp, C.c_double, C.c_double]),
    "sc_compare_fold_planes": (C.c_int, [_P, _dp, _dp, _dp, _dp]),
    "sc_compare_end": (C.c_int, [_P, _dp, _dp, _dp, _dp]),
    "sc_set_option": (C.c_int, [_P, C.c_char_p, C.c_double]),
    "sc_fill_nodata": (C.c_int, [_P, _dp, C.c_int, C.c_int, C.c_double, C.c_int,
                                 C.POINTER(C.c_longlong)]),
    "sc_curvature": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, _fp]),
    "sc_curvature_f64": (C.c_int, [_P] + [C.c_double] * 4 + [_dp]),
    "sc_get_near_ties": (C.c_int, [_P, _bp]),
    "sc_score_cells_f64": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int, C.c_int, _dp, _dp]),
    "sc_get_near_events": (C.c_int, [_P, _up, C.c_longlong, C.POINTER(C.c_longlong)]),
    "sc_score_pairs_f64": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, _dp, _dp]),
    "sc_settle_exact": (C.c_int, [_P, C.c_int, C.c_double, C.POINTER(C.c_longlong)]),
    "sc_snapshot_best": (C.c_int, [_P]),
    "sc_set_best": (C.c_int, [_P, _fp, _fp, _up]),
    "sc_rank_candidates": (C.c_int, [_P, _up, C.c_longlong, C.POINTER(C.c_longlong)]),
    "sc_settle_pairs": (C.c_int, [_P, C.POINTER(sc_template), C.c_int, _up, C.c_longlong, C.c_int, C.c_double,
                                  C.POINTER(C.c_longlong)]),
    "sc_exchange_candidates": (C.c_int, [_P, C.POINTER(C.c_longlong)]),
    "sc_get_resolution_stats": (C.c_int, [_P, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "sc_get_template_sums": (C.c_int, [_P, C.c_int, _dp, _dp]),
    "sc_profile": (C.c_int, [_P, C.c_int]),
    "sc_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_longlong), _dp]),
    "sc_kernel_name": (C.c_char_p, [C.c_int]),
    "sc_device_bytes": (C.c_size_t, [_P]),
    "sc_comm_unique_id": (C.c_int, [_P]),
    "sc_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "sc_gather_result": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int,
                                   _dp, _dp, C.c_int, _dp]),
    "sc_halo_exchange": (C.c_int, [_P, _dp] + [C.c_int] * 6
                         + [C.POINTER(sc_xfer), C.c_int, C.POINTER(_P)]),
    "sc_fold_ranks": (C.c_int, [_P]),
    "sc_comm_destroy": (C.c_int, [_P]),
    "sc_comm_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.c_char_p, C.c_int]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every exported function."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ScarpletHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as "
            "g; g.build()'` or `make -C scarplet_amd/csrc` (needs hipcc); this "
            "package has no CPU fallback" % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise ScarpletHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError = ABI mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.sc_abi_version() != ABI_VERSION:
        raise ScarpletHipError("libscarplet_hip.so ABI %d != expected %d"
                               % (lib.sc_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _as(arr, ptr_type):
    return arr.ctypes.data_as(ptr_type)


SPECTRA_MB = 8192.0


class Context(object):
    """One GPU's matcher state (an ``sc_ctx``)."""

    def __init__(self, device=0):
        self.lib = load()
        n = self.lib.sc_device_count()
        if n <= 0:
            raise ScarpletHipError(
                "no HIP device visible: scarplet_amd needs an AMD GPU "
                "(MI355X / gfx950); there is no CPU fallback")
        self._h = _P()
        rc = self.lib.sc_create(int(device), C.byref(self._h))
        if rc != SC_OK:
            raise ScarpletHipError("sc_create(device=%d) failed: %d"
                                   % (device, rc))
        self.device = int(device)
        self.core = None
        self.dem_key = None        # (geometry, cell size, the device's 128-bit fingerprint) of the resident block
        self.dem_nan = 0           # NaN cells the device found in it
        self.dem_unchanged = False # the last set_dem handed over what the context already held
        self.spectra_mb = 0.0
        # searches small enough for it keep their curvature spectra (sc_set_option "spectra_mb"): the
        # next search of the same DEM with the same tiles and orientations - the next scale of a
        # multi-scale job - starts from them
        self.set_option("spectra_mb", SPECTRA_MB)

    # -- plumbing ----------------------------------------------------------
    def _check(self, rc, what):
        if rc != SC_OK:
            msg = self.lib.sc_last_error(self._h)
            raise ScarpletHipError("%s failed (%d): %s" % (
                what, rc, msg.decode() if msg else ""))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.sc_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """Engine options (include/scarplet_hip.h sc_set_option): 'kappa',
        'variant', 'y_gb', 'spectra_mb'."""
        self._check(self.lib.sc_set_option(self._h, name.encode(), float(value)),
                    "sc_set_option(%s)" % name)
        if name == "spectra_mb":
            self.spectra_mb = float(value)

    def forget_spectra(self):
        """Drop the curvature spectra kept from earlier searches (option
        'spectra_mb'): the next search computes its own again."""
        self.set_option("spectra_mb", self.spectra_mb)

    def fill_nodata(self, z, max_search_distance, smoothing_iterations=0):
        """One GDALFillNodata-style pass over ``z`` (float64, NaN = nodata), in
        place; returns the number of cells still nodata."""
        assert z.dtype == np.float64 and z.flags.c_contiguous and z.ndim == 2
        left = C.c_longlong(0)
        self._check(self.lib.sc_fill_nodata(self._h, _as(z, _dp), z.shape[0], z.shape[1],
                                            float(max_search_distance), int(smoothing_iterations),
                                            C.byref(left)), "sc_fill_nodata")
        return int(left.value)

    # -- DEM ----------------------------------------------------------------
    def _dem_args(self, ly, lx, origin, shape, core, wrap):
        ny, nx = shape
        gy0, gx0 = origin
        if core is None:
            core = (0, ny, 0, nx)
        self.core = tuple(int(v) for v in core)
        self.shape = (int(ny), int(nx))
        return [int(ly), int(lx), int(gy0), int(gx0), int(ny), int(nx)] \
            + list(self.core)

    def set_dem(self, z, dx, dy, xaxis, yaxis, origin=(0, 0), shape=None,
                core=None, wrap=True):
        z = np.ascontiguousarray(z, dtype=np.float64)
        ly, lx = z.shape
        shape = z.shape if shape is None else shape
        xa = np.ascontiguousarray(xaxis, dtype=np.float64)
        ya = np.ascontiguousarray(yaxis, dtype=np.float64)
        assert xa.size == shape[1] and ya.size == shape[0]
        args = self._dem_args(ly, lx, origin, shape, core, wrap)
        self.dem_key = None
        self._check(self.lib.sc_set_dem(
            self._h, _as(z, _dp), *args, float(dx), float(dy), int(bool(wrap)),
            _as(xa, _dp), _as(ya, _dp)), "sc_set_dem")
        self._dem_info(args, dx, dy, wrap)

    def _dem_info(self, args, dx, dy, wrap):
        """What the device found in the block just handed over (sc_dem_info): NaN cells, the
        fingerprint, and whether it is the block the context already held."""
        nan, h2, same = C.c_longlong(0), (C.c_ulonglong * 2)(), C.c_int(0)
        self._check(self.lib.sc_dem_info(self._h, C.byref(nan), h2, C.byref(same)), "sc_dem_info")
        self.dem_nan, self.dem_unchanged = int(nan.value), bool(same.value)
        self.dem_key = (tuple(args), float(dx), float(dy), bool(wrap), int(h2[0]), int(h2[1]))

    def set_dem_device(self, z_dev, ly, lx, dx, dy, xaxis, yaxis, origin,
                       shape, core):
        xa = np.ascontiguousarray(xaxis, dtype=np.float64)
        ya = np.ascontiguousarray(yaxis, dtype=np.float64)
        args = self._dem_args(ly, lx, origin, shape, core, False)
        self.dem_key = None
        self._check(self.lib.sc_set_dem_device(
            self._h, z_dev, *args, float(dx), float(dy), 0, _as(xa, _dp),
            _as(ya, _dp)), "sc_set_dem_device")
        self._dem_info(args, dx, dy, False)

    def core_shape(self):
        cy0, cy1, cx0, cx1 = self.core
        return cy1 - cy0, cx1 - cx0

    def curvature(self, cc, sc2, ss, block_shape):
        out = np.empty(block_shape, dtype=np.float32)
        self._check(self.lib.sc_curvature(self._h, cc, sc2, ss, _as(out, _fp)),
                    "sc_curvature")
        return out

    def curvature_f64(self, alpha, block_shape):
        """dem.py:68-107 in float64 (sc_curvature_f64); cos / sin / squares by numpy, as the
        reference evaluates them."""
        out = np.empty(block_shape, dtype=np.float64)
        self._check(self.lib.sc_curvature_f64(self._h, float(np.cos(alpha) ** 2), float(np.sin(alpha)),
                                              float(np.cos(alpha)), float(np.sin(alpha) ** 2),
                                              _as(out, _dp)), "sc_curvature_f64")
        return out

    # -- generic plugin windows --------------------------------------------
    def upload_window(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        slot = C.c_int(-1)
        self._check(self.lib.sc_upload_window(
            self._h, _as(w, _dp), w.shape[0], w.shape[1], C.byref(slot)),
            "sc_upload_window")
        return slot.value

    def set_masks(self, slot, limits=None, err=None):
        def conv(m):
            if m is None:
                return None, None
            a = np.ascontiguousarray(m, dtype=np.uint8)
            return a, _as(a, _bp)
        la, lp = conv(limits)
        ea, ep = conv(err)
        self._check(self.lib.sc_set_masks(self._h, slot, lp, ep),
                    "sc_set_masks")

    def clear_windows(self):
        self._check(self.lib.sc_clear_windows(self._h), "sc_clear_windows")

    # -- hot path -----------------------------------------------------------
    def reset_best(self):
        self._check(self.lib.sc_reset_best(self._h), "sc_reset_best")

    def match(self, templates, plan, sync=True):
        fn = self.lib.sc_match if sync else self.lib.sc_match_async
        self._check(fn(self._h, templates, len(templates), C.byref(plan)),
                    "sc_match")

    def sync(self):
        self._check(self.lib.sc_sync(self._h), "sc_sync")

    def match_template(self, template, plan):
        h, w = self.core_shape()
        amp = np.empty((h, w), dtype=np.float32)
        snr = np.empty((h, w), dtype=np.float32)
        self._check(self.lib.sc_match_template(
            self._h, C.byref(template), C.byref(plan), _as(amp, _fp),
            _as(snr, _fp)), "sc_match_template")
        return amp, snr

    def get_best(self):
        h, w = self.core_shape()
        amp = np.empty((h, w), dtype=np.float32)
        snr = np.empty((h, w), dtype=np.float32)
        idx = np.empty((h, w), dtype=np.uint32)
        self._check(self.lib.sc_get_best(self._h, _as(amp, _fp), _as(snr, _fp),
                                         _as(idx, _up)), "sc_get_best")
        return amp, snr, idx

    def get_result(self, param_of_id, angle_of_id):
        """(4, h, w) float64: amp, age, angle, snr of the running best."""
        h, w = self.core_shape()
        par = np.ascontiguousarray(param_of_id, dtype=np.float64)
        ang = np.ascontiguousarray(angle_of_id, dtype=np.float64)
        # (large results are views of recycled host blocks, already faulted in: _hostpool)
        from scarplet_amd import _hostpool
        out = _hostpool.empty((4, h, w), dtype=np.float64)
        self._check(self.lib.sc_get_result(self._h, _as(par, _dp), _as(ang, _dp),
                                           len(par), _as(out, _dp)), "sc_get_result")
        return out

    def template_sums(self, n):
        a = np.empty(n)
        b = np.empty(n)
        self._check(self.lib.sc_get_template_sums(self._h, n, _as(a, _dp),
                                                  _as(b, _dp)),
                    "sc_get_template_sums")
        return a, b

    # -- measurement ----------------------------------------------------------
    def profile(self, stride):
        self._check(self.lib.sc_profile(self._h, int(stride)), "sc_profile")

    def profile_get(self):
        out = {}
        for k, name in enumerate(K_NAMES):
            n = C.c_longlong(0)
            ms = C.c_double(0)
            self._check(self.lib.sc_profile_get(self._h, k, C.byref(n),
                                                C.byref(ms)), "sc_profile_get")
            out[name] = (n.value, ms.value)
        return out

    def device_bytes(self):
        return int(self.lib.sc_device_bytes(self._h))

    # -- multi-GPU ------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(COMM_ID_BYTES)
        rc = self.lib.sc_comm_unique_id(buf)
        if rc != SC_OK:
            raise ScarpletHipError("sc_comm_unique_id failed: %d" % rc)
        return buf.raw

    def comm_init(self, uid, rank, nranks):
        buf = C.create_string_buffer(bytes(uid), COMM_ID_BYTES)
        self._check(self.lib.sc_comm_init(self._h, buf, rank, nranks),
                    "sc_comm_init")

    def near_ties(self):
        """(h, w) uint8: 1 where an FFT search since the last reset saw a near-tie (option "near_window")."""
        h, w = self.core_shape()
        out = np.zeros((h, w), dtype=np.uint8)
        self._check(self.lib.sc_get_near_ties(self._h, _as(out, _bp)), "sc_get_near_ties")
        return out

    def score_cells_f64(self, cells, n_templates):
        """(amp, snr) float64, each (m, n_templates): match_template() in float64 at the m global cells (rows of
        (i, j)) for every template of the last match in this context, in hand-over order (sc_score_cells_f64)."""
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 2)
        m = len(cells)
        amp = np.empty((m, int(n_templates)), dtype=np.float64)
        snr = np.empty((m, int(n_templates)), dtype=np.float64)
        self._check(self.lib.sc_score_cells_f64(self._h, cells.ctypes.data_as(C.POINTER(C.c_int32)), m, int(n_templates),
                                                _as(amp, _dp), _as(snr, _dp)), "sc_score_cells_f64")
        return amp, snr

    def near_events(self):
        """The near-ties of the searches since the last reset as events, (n, 4) uint32: cell (row-major index into the
        core planes), id of the template scored, id of the record's holder at that moment, float32 bits of the larger of
        their two scores (sc_get_near_events) - or None where the device list overflowed (more events than two per core cell)."""
        n = C.c_longlong(0)
        try:
            self._check(self.lib.sc_get_near_events(self._h, None, 0, C.byref(n)), "sc_get_near_events")
            want = int(n.value)
            if want == 0:
                return np.zeros((0, 4), dtype=np.uint32)
            ev = np.empty((want, 4), dtype=np.uint32)
            self._check(self.lib.sc_get_near_events(self._h, _as(ev, _up), want, C.byref(n)), "sc_get_near_events")
        except ScarpletHipError as e:
            if "overflowed" in str(e):                   # (the library says so itself since ABI 8)
                return None
            raise
        if int(n.value) != want:
            return None
        return ev

    def score_pairs_f64(self, cells, templates):
        """(amp, snr) float64, each (m,): match_template() in float64 at global cell cells[k] for template templates[k]
        (index in the last match's hand-over order) - sc_score_pairs_f64."""
        cells = np.ascontiguousarray(cells, dtype=np.int32).reshape(-1, 2)
        templates = np.ascontiguousarray(templates, dtype=np.int32).reshape(-1)
        m = len(cells)
        amp = np.empty(m, dtype=np.float64)
        snr = np.empty(m, dtype=np.float64)
        i32 = C.POINTER(C.c_int32)
        self._check(self.lib.sc_score_pairs_f64(self._h, cells.ctypes.data_as(i32), templates.ctypes.data_as(i32), m,
                                                _as(amp, _dp), _as(snr, _dp)), "sc_score_pairs_f64")
        return amp, snr

    def settle_exact(self, n_twin=0, max_work=0.0):
        """exact=True on the device (sc_settle_exact): the near-tie cells of the last search (option "near_window") take
        their float64 argmax among the templates their events name.  Returns the counters as a dict."""
        st = (C.c_longlong * 8)()
        self._check(self.lib.sc_settle_exact(self._h, int(n_twin), float(max_work), st), "sc_settle_exact")
        return self._settle_stats(st)

    @staticmethod
    def _settle_stats(st):
        return {"flagged_cells": int(st[0]), "pairs_listed": int(st[1]), "float64_pairs": int(st[2]),
                "float64_cells": int(st[3]), "changed_cells": int(st[4]), "events": int(st[5]),
                "taps": int(st[7])}

    # ---- exact mode of an orientation-sharded search (include/scarplet_hip.h: sc_settle_pairs) ----
    def snapshot_best(self):
        """Keep the record's (snr, id) planes as they stand on the device - before the fold over the ranks."""
        self._check(self.lib.sc_snapshot_best(self._h), "sc_snapshot_best")

    def set_best(self, amp, snr, idx):
        """Upload a record (core-shaped float32, float32, uint32) as the running best (sc_set_best)."""
        h, w = self.core_shape()
        amp = np.ascontiguousarray(amp, dtype=np.float32)
        snr = np.ascontiguousarray(snr, dtype=np.float32)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        if not (amp.shape == snr.shape == idx.shape == (h, w)):
            raise ValueError("set_best: the record must have the core's shape %r" % ((h, w),))
        self._check(self.lib.sc_set_best(self._h, _as(amp, _fp), _as(snr, _fp), _as(idx, _up)), "sc_set_best")

    def rank_candidates(self, fetch=True):
        """This rank's candidates against the folded record: an (n, 2) uint32 array of (core cell index, template id)
        (sc_rank_candidates: the templates of this rank's events and its own holder that lie within the near-tie window
        of the record as it stands now).  ``fetch=False``: the list stays on the device (exchange_candidates), the count
        is returned."""
        n = C.c_longlong(0)
        self._check(self.lib.sc_rank_candidates(self._h, None, 0, C.byref(n)), "sc_rank_candidates")
        if not fetch:
            return n.value
        out = np.empty((n.value, 2), dtype=np.uint32)
        if n.value:
            m = C.c_longlong(0)
            self._check(self.lib.sc_rank_candidates(self._h, _as(out, _up), n.value, C.byref(m)), "sc_rank_candidates")
            if m.value != n.value:
                raise ScarpletHipError("sc_rank_candidates: %d pairs, then %d" % (n.value, m.value))
        return out

    def exchange_candidates(self):
        """All ranks' candidate lists as one list on every device (sc_exchange_candidates: two all-gathers over RCCL);
        returns its length in pairs, padding included - what settle_pairs(templates, None, ...) then settles."""
        n = C.c_longlong(0)
        self._check(self.lib.sc_exchange_candidates(self._h, C.byref(n)), "sc_exchange_candidates")
        self._exchanged = n.value
        return n.value

    def settle_pairs(self, templates, pairs, n_twin=0, max_work=0.0):
        """Settle the union of all ranks' candidates with the descriptors of the WHOLE search (sc_settle_pairs); the
        counters as settle_exact returns them.  ``pairs``: an (n, 2) uint32 array, or None for the list
        exchange_candidates left on the device."""
        st = (C.c_longlong * 8)()
        if pairs is None:
            ptr, n = None, int(getattr(self, "_exchanged", 0))
        else:
            pairs = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
            ptr, n = (_as(pairs, _up) if len(pairs) else None), len(pairs)
        self._check(self.lib.sc_settle_pairs(self._h, templates, len(templates), ptr, n, int(n_twin), float(max_work), st),
                    "sc_settle_pairs")
        return self._settle_stats(st)

    def comm_destroy(self):
        """Drop this context's RCCL communicator (sc_comm_destroy); nothing to do without one."""
        self._check(self.lib.sc_comm_destroy(self._h), "sc_comm_destroy")

    def resolution_stats(self):
        """(wins, wins near the float32 resolution floor) of the FFT searches since the last
        reset_best (sc_get_resolution_stats)."""
        a, b = C.c_longlong(0), C.c_longlong(0)
        self._check(self.lib.sc_get_resolution_stats(self._h, C.byref(a), C.byref(b)), "sc_get_resolution_stats")
        return a.value, b.value

    def comm_info(self):
        """What RCCL reports for this context's communicator (sc_comm_info): a dict with
        nranks (0: no communicator), rank, device and the device's PCI bus id."""
        n, r, d = C.c_int(0), C.c_int(-1), C.c_int(-1)
        bus = C.create_string_buffer(32)
        self._check(self.lib.sc_comm_info(self._h, C.byref(n), C.byref(r), C.byref(d), bus, 32), "sc_comm_info")
        return {"nranks": n.value, "rank": r.value, "device": d.value, "bus_id": bus.value.decode()}

    def fold_ranks(self):
        """Collective fold of the ranks' running-best records (orientation-sharded search)."""
        self._check(self.lib.sc_fold_ranks(self._h), "sc_fold_ranks")

    def gather_result(self, root, cores, shape, param_of_id, angle_of_id, is_root, out=None):
        """Collective final gather over RCCL; returns (4, ny, nx) on root (``out``: a float64
        array of that shape to fill instead of a new one - the cores tile the DEM, every cell
        is written)."""
        cores = np.ascontiguousarray(cores, dtype=np.int32)
        par = np.ascontiguousarray(param_of_id, dtype=np.float64)
        ang = np.ascontiguousarray(angle_of_id, dtype=np.float64)
        if is_root and out is None:
            out = np.zeros((4,) + tuple(shape), dtype=np.float64)
        if is_root:
            assert out.dtype == np.float64 and out.flags.c_contiguous and out.shape == (4,) + tuple(shape)
        self._check(self.lib.sc_gather_result(
            self._h, int(root), cores.ctypes.data_as(C.POINTER(C.c_int32)), int(shape[0]),
            int(shape[1]), _as(par, _dp), _as(ang, _dp), len(par),
            _as(out, _dp) if is_root else None), "sc_gather_result")
        return out

    def halo_exchange(self, core, halo, xfers):
        core = np.ascontiguousarray(core, dtype=np.float64)
        arr = (sc_xfer * max(len(xfers), 1))()
        for i, x in enumerate(xfers):
            arr[i] = sc_xfer(*[int(v) for v in x])
        z_dev = _P()
        self._check(self.lib.sc_halo_exchange(
            self._h, _as(core, _dp), core.shape[0], core.shape[1],
            int(halo[0]), int(halo[1]), int(halo[2]), int(halo[3]), arr,
            len(xfers), C.byref(z_dev)), "sc_halo_exchange")
        return z_dev
