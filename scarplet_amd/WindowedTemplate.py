"""Template plugins for windowed template matching (host side).

Drop-in for the reference's ``scarplet/WindowedTemplate.py`` plugin API: a
template class is callable as ``Template(scale, age, angle, nx, ny, de)`` and
offers ``template()``, ``get_window_limits()`` and optionally
``get_err_mask()`` (the contract ``match_template`` relies on,
reference core.py:345-346, 369-375).

The numpy methods here exist for API compatibility and for user code that
inspects templates; the matcher itself never calls them for the built-in
classes.  Built-ins additionally describe themselves to the HIP library
through ``_device_descriptor()``: a handful of float64 scalars (evaluated
with numpy exactly as the reference evaluates them, so that the support
``W != 0`` decided on the device in float64 is the reference's support) from
which the kernels synthesise W, the window and the masks on the GPU.  Any
other subclass of ``WindowedTemplate`` is handled generically: its numpy
``template()`` is evaluated on the host and uploaded (see core.py).

Reference lines are cited as WT.py:<line> (= scarplet/WindowedTemplate.py).
"""

import bisect

import numpy as np
from scipy.special import erfinv

# kinds understood by the device library (include/scarplet_hip.h)
KIND_SCARP = 0
KIND_RICKER = 1
KIND_WINDOW = 2          # explicit W window uploaded from the host

FLAG_NEGATE = 1          # W -> -W (RightFacingUpperBreakScarp, WT.py:254-255)
FLAG_ERR_XR_LE0 = 2      # snr = 0 where xr <= 0 (WT.py:265-267)
FLAG_ERR_XR_GE0 = 4      # snr = 0 where xr >= 0 (WT.py:302-304)
FLAG_NO_LIMITS = 8       # get_window_limits() is all-False (WT.py:495-496)

# float64 exp(-u) is non-zero iff u < 1075*ln(2); this is what bounds the
# support of a Ricker template (W != 0), SURVEY.md section 7.
EXP_UNDERFLOW = 745.1332191019412


_AXIS_CACHE = {}
_AXIS_LIST_CACHE = {}
_TRIG_CACHE = {}
_ERFINV_09 = erfinv(0.9)          # WT.py:156


def _trig(alpha):
    """(cos a, sin a, cos(a - pi/2), sin(a - pi/2)) evaluated with numpy as the
    reference evaluates them (WT.py:55-56, 68-73), cached: a search builds 35
    templates per orientation."""
    t = _TRIG_CACHE.get(alpha)
    if t is None:
        t = (np.cos(alpha), np.sin(alpha), np.cos(alpha - np.pi / 2), np.sin(alpha - np.pi / 2))
        if len(_TRIG_CACHE) > 4096:
            _TRIG_CACHE.clear()
        _TRIG_CACHE[alpha] = t
    return t


def _axis_list(n, de):
    """centred_axis as a Python list (for bisect), None when it is not ascending."""
    key = (int(n), float(de))
    if key not in _AXIS_LIST_CACHE:
        a = centred_axis(n, de)
        if len(_AXIS_LIST_CACHE) > 64:
            _AXIS_LIST_CACHE.clear()
        _AXIS_LIST_CACHE[key] = a.tolist() if a[0] <= a[-1] else None
    return _AXIS_LIST_CACHE[key]


def centred_axis(n, de):
    """Cell-centre coordinates of one grid axis, mean removed (WT.py:50-53).

    Evaluated with the same numpy operations as the reference so that the
    values (and therefore every ``< c`` / ``!= 0`` decision taken on them)
    are bit-identical.  Cached: a search builds thousands of templates on the
    same grid."""
    key = (int(n), float(de))
    a = _AXIS_CACHE.get(key)
    if a is None:
        a = de * np.linspace(1, n, num=n)
        a = a - np.mean(a)
        a.setflags(write=False)
        if len(_AXIS_CACHE) > 64:
            _AXIS_CACHE.clear()
        _AXIS_CACHE[key] = a
    return a


def _interval_le(a, lo, hi, alist=None):
    """Indices [i0, i1] of the cells of the monotonic axis ``a`` with
    lo <= a <= hi, found by bisection (same float compares as the elementwise
    test, O(log n)).  ``alist``: the same axis as a Python list (_axis_list)."""
    if alist is not None:
        return bisect.bisect_left(alist, lo), bisect.bisect_right(alist, hi) - 1
    if a[0] <= a[-1]:
        i0 = int(np.searchsorted(a, lo, side="left"))
        i1 = int(np.searchsorted(a, hi, side="right")) - 1
    else:
        keep = np.nonzero((a >= lo) & (a <= hi))[0]
        if keep.size == 0:
            return 0, -1
        i0, i1 = int(keep[0]), int(keep[-1])
    return i0, i1


class WindowedTemplate(object):
    """Base class: rotated rectangular window of half-widths ``c`` (along xr)
    and ``d`` (along yr) on an ``ny`` x ``nx`` grid of spacing ``de``."""

    d = None
    alpha = None
    c = None
    nx = None
    ny = None
    de = None

    def _axes(self):
        return centred_axis(self.nx, self.de), centred_axis(self.ny, self.de)

    def get_coordinates(self):
        """Rotated coordinates (xr, yr) of every grid cell (WT.py:49-59)."""
        x, y = self._axes()
        x = x[np.newaxis, :]
        y = y[:, np.newaxis]
        ca, sa = np.cos(self.alpha), np.sin(self.alpha)
        return x * ca + y * sa, -x * sa + y * ca

    def get_mask(self):
        """Cells inside the template window (WT.py:61-64)."""
        xr, yr = self.get_coordinates()
        return (abs(xr) < self.c) & (abs(yr) < self.d)

    def _limit_margins(self):
        """(an_x, an_y) of WT.py:68-73."""
        d, c = self.d, self.c
        ca, sa, cam, sam = _trig(self.alpha)
        x4 = d * cam
        y4 = d * sam
        x1 = d * ca
        y1 = d * sa
        an_y = abs((x4 - x1) + 2 * c * cam)
        an_x = abs((y1 - y4) + 2 * c * sam)
        return an_x, an_y

    def _limit_axes(self):
        """1-D form of get_window_limits(): boolean 'masked' vectors for the
        columns and rows (WT.py:75-82)."""
        an_x, an_y = self._limit_margins()
        x, y = self._axes()
        mx = (x < (min(x) + an_x)) | (x > (max(x) - an_x))
        my = (y < (min(y) + an_y)) | (y > (max(y) - an_y))
        return mx, my

    def get_window_limits(self):
        """Mask of cells too close to the grid edge for the window to fit
        (WT.py:66-84)."""
        mx, my = self._limit_axes()
        return mx[np.newaxis, :] | my[:, np.newaxis]

    # -- device description -------------------------------------------------
    def _kept_bounds(self):
        """Window limits as index bounds: a cell (i, j) is kept iff
        ilo <= i <= ihi and jlo <= j <= jhi (empty when lo > hi).  Same float
        compares as get_window_limits(): masked means x < min(x) + an_x or
        x > max(x) - an_x (WT.py:81-82)."""
        an_x, an_y = self._limit_margins()
        x, y = self._axes()
        jlo, jhi = _interval_le(x, min(x[0], x[-1]) + an_x, max(x[0], x[-1]) - an_x,
                                _axis_list(self.nx, self.de))
        ilo, ihi = _interval_le(y, min(y[0], y[-1]) + an_y, max(y[0], y[-1]) - an_y,
                                _axis_list(self.ny, self.de))
        return ilo, ihi, jlo, jhi

    def _support_bbox(self, c_eff=None):
        """Conservative bounding box of the support in centred offsets
        p = k - ny//2, q = l - nx//2 (clipped to the grid)."""
        c = self.c if c_eff is None else min(self.c, c_eff)
        t = _trig(self.alpha)
        ca, sa = abs(t[0]), abs(t[1])
        bx = (c * ca + self.d * sa) * (1 + 1e-12) + 1e-300
        by = (c * sa + self.d * ca) * (1 + 1e-12) + 1e-300
        x, y = self._axes()
        l0, l1 = _interval_le(x, -bx, bx, _axis_list(self.nx, self.de))
        k0, k1 = _interval_le(y, -by, by, _axis_list(self.ny, self.de))
        return (k0 - self.ny // 2, k1 - self.ny // 2,
                l0 - self.nx // 2, l1 - self.nx // 2)


class Scarp(WindowedTemplate):
    """Curvature template of a vertical scarp of morphologic age ``kt``
    (Hanks 2000; Hilley et al. 2010), WT.py:87-183."""

    _kind = KIND_SCARP
    _flags = 0

    def __init__(self, d, kt, alpha, nx, ny, de):
        self.d = d
        self.kt = kt
        self.alpha = -alpha                        # WT.py:151
        self.nx = nx
        self.ny = ny
        self.de = de
        self.c = abs(2 * np.sqrt(self.kt) * _ERFINV_09)    # WT.py:156-157

    def _profile(self, xr):
        return (-xr / (2. * self.kt ** (3 / 2.) * np.sqrt(np.pi))) \
            * np.exp(-xr ** 2. / (4. * self.kt))

    def template(self):
        """W = profile(xr) inside the window, 0 outside (WT.py:159-183)."""
        xr, yr = self.get_coordinates()
        inside = (abs(xr) < self.c) & (abs(yr) < self.d)
        return self._profile(xr) * inside

    def template_numexpr(self):
        """Same values as template(); kept for API compatibility
        (WT.py:185-215 differs from template() only in the evaluator)."""
        return self.template()

    def _device_descriptor(self):
        kt = self.kt
        t = _trig(self.alpha)
        return dict(kind=self._kind, flags=self._flags,
                    cos_a=float(t[0]), sin_a=float(t[1]),
                    c=float(self.c), d=float(self.d),
                    p0=float(2. * kt ** (3 / 2.) * np.sqrt(np.pi)),
                    p1=float(4. * kt),
                    limits=self._kept_bounds(), bbox=self._support_bbox())


class RightFacingUpperBreakScarp(Scarp):
    """Upper slope break of a right-facing scarp: sign-flipped template and
    an error mask over the lower half (WT.py:218-267)."""

    _flags = FLAG_NEGATE | FLAG_ERR_XR_LE0

    def template(self):
        return -Scarp.template(self)

    def get_err_mask(self):
        xr, _ = self.get_coordinates()
        return xr <= 0


class LeftFacingUpperBreakScarp(Scarp):
    """Upper slope break of a left-facing scarp (WT.py:270-304)."""

    _flags = FLAG_ERR_XR_GE0

    def get_err_mask(self):
        xr, _ = self.get_coordinates()
        return xr >= 0


class ShiftedTemplateMixin(WindowedTemplate):
    """Template offset from the window centre by (dx, dy) cells
    (WT.py:307-421).  Only reachable through
    calculate_best_fit_parameters_serial, which forwards **kwargs."""

    _kind = KIND_WINDOW

    def __init__(self, *args, **kwargs):
        super().__init__(*args)
        self.set_offset(kwargs['dx'], kwargs['dy'])

    def set_offset(self, dx, dy):
        self.dx = dx
        self.dy = dy

    def shift_template(self, W, dx, dy):
        """Shift W right by dx and down by dy cells, zero filling
        (WT.py:368-408; dx <= 0 / dy <= 0 shift the other way)."""
        ny, nx = W.shape
        out = np.zeros_like(W)
        if dx > 0:
            out[:, dx:] = W[:, :nx - dx]
        else:
            out[:, :nx + dx] = W[:, -dx:]
        W, out = out, np.zeros_like(W)
        if dy > 0:
            # reference quirk kept: dy > 0 drops the LAST dy rows and pads
            # at the bottom, i.e. the content does not move (WT.py:398-401)
            out[:ny - dy, :] = W[:ny - dy, :]
        else:
            out[-dy:, :] = W[-dy:, :]           # WT.py:403-406, likewise
        return out

    def template(self):
        W = super().template()
        return self.shift_template(W, self.dx, self.dy)

    def _device_descriptor(self):
        return None                              # generic (host window) path


class ShiftedLeftFacingUpperBreakScarp(ShiftedTemplateMixin,
                                       LeftFacingUpperBreakScarp):
    pass


class ShiftedRightFacingUpperBreakScarp(ShiftedTemplateMixin,
                                        RightFacingUpperBreakScarp):
    pass


class Ricker(WindowedTemplate):
    """2-D Ricker wavelet of frequency ``f`` (Lashermes et al. 2007),
    WT.py:434-520.  The second constructor argument (the matcher's ``age``)
    is the wavelet frequency."""

    _kind = KIND_RICKER
    _flags = FLAG_NO_LIMITS

    def __init__(self, d, f, alpha, nx, ny, de):
        self.d = d
        self.f = f
        self.alpha = -alpha                        # WT.py:489
        self.nx = nx
        self.ny = ny
        self.c = nx                                # WT.py:492
        self.de = de

    def get_window_limits(self):
        return np.zeros((self.ny, self.nx), dtype=bool)     # WT.py:495-496

    def template(self):
        xr, yr = self.get_coordinates()
        u2 = (np.pi * self.f * xr) ** 2.
        W = (1. - 2. * u2) * np.exp(-u2)
        return W * ((abs(xr) < self.c) & (abs(yr) < self.d))

    def _device_descriptor(self):
        pif = float(np.pi * self.f)
        c_eff = np.sqrt(EXP_UNDERFLOW) / abs(pif) if pif != 0 else np.inf
        t = _trig(self.alpha)
        return dict(kind=self._kind, flags=self._flags,
                    cos_a=float(t[0]), sin_a=float(t[1]),
                    c=float(self.c), d=float(self.d), p0=pif, p1=0.0,
                    limits=(0, self.ny - 1, 0, self.nx - 1),
                    bbox=self._support_bbox(c_eff=c_eff))


class Channel(Ricker):
    """Ricker wavelet used for fluvial channels (WT.py:523-525)."""
    pass


# ---- the reference's own built-in classes ------------------------------------------------------
_TWIN_CACHE = {}
# probe grids of builtin_twin: (scale d, second argument, orientation, nx, ny, de) - odd and even
# sizes, a non-unit cell size, the three orientations at which a Scarp window has a zero column
_TWIN_PROBES = [(6.0, 0.7, -1.1, 23, 18, 1.5), (5.0, 3.0, 0.4, 20, 24, 1.0), (4.0, 1.3, np.pi / 2, 17, 17, 1.0),
                (7.0, 0.15, 0.0, 26, 21, 2.0), (5.0, 0.4, -np.pi / 2, 19, 22, 1.0)]


def _same_template_behaviour(Theirs, Ours):
    """True when ``Theirs`` answers the plugin contract (core.py:345-346, 369-375) exactly as this
    package's class ``Ours`` does on the probe grids: the same window half-widths and alpha, the
    same support ``W != 0`` cell for cell, the same window-limit and error masks in every cell, and
    W equal to 1e-13 of its largest value (the reference's Ricker and UpperBreak classes evaluate
    through numexpr, WT.py:205-213, 514-515, whose exp may differ from numpy's in the last place;
    the device synthesises W in its own float64 arithmetic from the descriptor either way)."""
    for d, par, ang, nx, ny, de in _TWIN_PROBES:
        a, b = Theirs(d, par, ang, nx, ny, de), Ours(d, par, ang, nx, ny, de)
        for k in ("c", "d", "alpha", "nx", "ny", "de"):
            if getattr(a, k, None) != getattr(b, k):
                return False
        wa, wb = np.asarray(a.template(), dtype=np.float64), np.asarray(b.template(), dtype=np.float64)
        if wa.shape != wb.shape or not np.array_equal(wa != 0, wb != 0):
            return False
        if not np.all(np.abs(wa - wb) <= 1e-13 * np.abs(wb).max()):
            return False
        if not np.array_equal(np.asarray(a.get_window_limits(), dtype=bool), b.get_window_limits()):
            return False
        if hasattr(a, "get_err_mask") != hasattr(b, "get_err_mask"):
            return False
        if hasattr(a, "get_err_mask") and not np.array_equal(np.asarray(a.get_err_mask(), dtype=bool),
                                                              b.get_err_mask()):
            return False
    return True


def builtin_twin(Template):
    """This package's built-in class that ``Template`` stands for, or None.

    A user script written against the reference does ``from scarplet.WindowedTemplate import
    Scarp`` (WT.py:87-215, 434-525) and hands THAT class to ``match``.  It has no
    ``_device_descriptor`` and would take the generic plugin path - a full-grid numpy
    ``template()`` and two full-grid masks per (age, orientation).  A class is taken for one of
    the built-ins when it carries a built-in's name, lives in a module called
    ``WindowedTemplate``, describes nothing to the device itself, and behaves like this package's
    class of that name on five small probe grids (_same_template_behaviour: supports and masks
    cell for cell).  Anything else - a subclass, a renamed or edited copy - stays on the generic
    path, which evaluates whatever the class computes.  The verdict is cached per class object."""
    own = (Scarp, RightFacingUpperBreakScarp, LeftFacingUpperBreakScarp, Ricker, Channel)
    if Template in own:
        return Template
    try:
        return _TWIN_CACHE[Template]
    except (KeyError, TypeError):
        pass
    twin = None
    name = getattr(Template, "__name__", None)
    mod = str(getattr(Template, "__module__", "")).rsplit(".", 1)[-1]
    cand = {c.__name__: c for c in own}.get(name)
    if cand is not None and mod == "WindowedTemplate" and not hasattr(Template, "_device_descriptor"):
        try:
            if _same_template_behaviour(Template, cand):
                twin = cand
        except Exception:
            twin = None
    try:
        _TWIN_CACHE[Template] = twin
    except TypeError:
        pass
    return twin


# ---- descriptors of a whole (angle, parameter) grid at once ------------------------------------
def grid_descriptors(Template, scale, params, angles, nx, ny, de):
    """``_device_descriptor()`` of ``Template(scale, p, a, nx, ny, de)`` for every angle a and
    parameter p, as arrays of shape (n_angles, n_params) - or None when ``Template`` is neither
    exactly one of the built-in classes (a subclass may override anything) nor the reference's own
    class of that name (builtin_twin), or the grid axes are not ascending.  A search builds thousands of descriptors; one Python object and three dozen
    numpy scalar calls each cost more than the device spends on a small search.

    Bit-identical to the per-template path: every transcendental (cos, sin, sqrt, pow) is still
    evaluated per scalar exactly as there; only the additions, multiplications and the bisection
    are done on arrays (tests/test_templates.py compares the two exhaustively)."""
    Template = builtin_twin(Template)
    if Template is None:
        return None
    scarp_like = Template in (Scarp, RightFacingUpperBreakScarp, LeftFacingUpperBreakScarp)
    ricker_like = Template in (Ricker, Channel)
    x, y = centred_axis(nx, de), centred_axis(ny, de)
    if not (x[0] <= x[-1] and y[0] <= y[-1]):
        return None
    params = [p for p in params]
    na, npar = len(angles), len(params)
    trig = [_trig(-a) for a in angles]                 # alpha = -angle (WT.py:151, 489)
    col = lambda k: np.array([t[k] for t in trig], dtype=np.float64)[:, np.newaxis]
    ca, sa, cam, sam = col(0), col(1), col(2), col(3)
    d = scale
    if scarp_like:
        c = np.array([abs(2 * np.sqrt(kt) * _ERFINV_09) for kt in params], dtype=np.float64)[np.newaxis, :]
        p0 = np.array([float(2. * kt ** (3 / 2.) * np.sqrt(np.pi)) for kt in params])[np.newaxis, :]
        p1 = np.array([float(4. * kt) for kt in params])[np.newaxis, :]
        c_box = c
        # window limits (WT.py:68-82)
        x4, y4, x1, y1 = d * cam, d * sam, d * ca, d * sa
        an_y = abs((x4 - x1) + 2 * c * cam)
        an_x = abs((y1 - y4) + 2 * c * sam)
        jlo = np.searchsorted(x, min(x[0], x[-1]) + an_x, side="left")
        jhi = np.searchsorted(x, max(x[0], x[-1]) - an_x, side="right") - 1
        ilo = np.searchsorted(y, min(y[0], y[-1]) + an_y, side="left")
        ihi = np.searchsorted(y, max(y[0], y[-1]) - an_y, side="right") - 1
    else:
        pif = [float(np.pi * f) for f in params]
        c = np.full((1, npar), float(nx))              # WT.py:492
        c_eff = np.array([np.sqrt(EXP_UNDERFLOW) / abs(v) if v != 0 else np.inf for v in pif])[np.newaxis, :]
        c_box = np.minimum(c, c_eff)
        p0 = np.array(pif)[np.newaxis, :]
        p1 = np.zeros((1, npar))
        ilo = np.zeros((na, npar), dtype=np.int64)
        ihi = np.full((na, npar), ny - 1, dtype=np.int64)
        jlo = np.zeros((na, npar), dtype=np.int64)
        jhi = np.full((na, npar), nx - 1, dtype=np.int64)
    # support box (_support_bbox)
    aca, asa = abs(ca), abs(sa)
    bx = (c_box * aca + d * asa) * (1 + 1e-12) + 1e-300
    by = (c_box * asa + d * aca) * (1 + 1e-12) + 1e-300
    l0 = np.searchsorted(x, -bx, side="left")
    l1 = np.searchsorted(x, bx, side="right") - 1
    k0 = np.searchsorted(y, -by, side="left")
    k1 = np.searchsorted(y, by, side="right") - 1
    full = lambda v: np.broadcast_to(np.asarray(v, dtype=np.float64), (na, npar))
    return dict(kind=Template._kind, flags=Template._flags, cos_a=full(ca), sin_a=full(sa),
                c=full(c), d=full(float(d)), p0=full(p0), p1=full(p1),
                ilo=ilo, ihi=ihi, jlo=jlo, jhi=jhi,
                pmin=k0 - ny // 2, pmax=k1 - ny // 2, qmin=l0 - nx // 2, qmax=l1 - nx // 2)
