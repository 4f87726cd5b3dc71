"""Host-side planning for the device matcher: search grids, template
descriptors and the overlap-save tiling of the FFT path.

Pure numpy; no device access here, so this is what the CPU test-suite
exercises (tests/test_plan.py checks the tiling against the oracle through a
numpy model of the kernels).

Index conventions (verified against the reference in oracle/gen_golden.py):

    xcorr[i, j] = sum_{p,q} W[ny//2 + p, nx//2 + q]
                      * curv[(i - p + oy) % ny, (j - q + ox) % nx]

with oy = ny % 2, ox = nx % 2: the reference's fft2/ifft2/fftshift sequence
(core.py:349-363) is a circular convolution with the template centred at
(ny//2, nx//2); for odd sizes the centre pixel and the fftshift disagree by
one cell.  T3 is the same sum with W -> (W != 0) and curv -> curv**2.
"""

import math

import numpy as np

METHOD_DIRECT = 0
METHOD_FFT = 1
METHOD_AUTO = 2

T_MIN = 64
T_MAX = 4096


def angle_grid(ang_min=-np.pi / 2, ang_max=np.pi / 2):
    """Orientation grid of the reference search driver (core.py:173-175):
    one-degree steps, both ends included."""
    num = int((180 / np.pi) * (ang_max - ang_min) / 1 + 1)
    return np.linspace(ang_min, ang_max, num)


def age_grid():
    """Morphologic-age grid of the reference (core.py:107, 286): 35 values,
    10**0 ... 10**3.4 m2."""
    return 10 ** np.arange(0, 3.5, 0.1)


def curvature_coefficients(angle):
    """(cc, sc2, ss) with curv = cc*A - sc2*B + ss*C (dem.py:103-104)."""
    s, c = np.sin(angle), np.cos(angle)
    return float(c ** 2), float(2 * s * c), float(s ** 2)


def bbox_union(bboxes):
    """Union of (pmin, pmax, qmin, qmax) boxes; empty boxes are skipped."""
    pmin = qmin = 0
    pmax = qmax = 0
    for (a, b, c, d) in bboxes:
        if b >= a:
            pmin, pmax = min(pmin, a), max(pmax, b)
        if d >= c:
            qmin, qmax = min(qmin, c), max(qmax, d)
    return pmin, pmax, qmin, qmax


def halo_for(bbox, ny, nx):
    """Rows/columns of DEM a tile needs beyond its core on each side, for a
    support box in centred offsets: (lo_y, hi_y, lo_x, hi_x).  Output pixel i
    reads curvature rows i - pmax + oy ... i - pmin + oy."""
    pmin, pmax, qmin, qmax = bbox
    oy, ox = ny % 2, nx % 2
    return (max(pmax - oy, 0), max(-pmin + oy, 0),
            max(qmax - ox, 0), max(-qmin + ox, 0))


def _is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


# Relative cost per padded cell of the FFT kernels by tile size, measured on
# MI355X (tools/crossover.py): the kernels are built around 512 ..
# 2048 (one 16-point set per thread, 128..512 threads per workgroup); smaller
# tiles leave most of a workgroup idle, 4096 runs the generic two-set path.
TILE_PENALTY = {64: 20.0, 128: 8.0, 256: 5.0, 512: 1.0, 1024: 1.0, 2048: 1.0,
                4096: 2.3}


# The same along y (column length of the inverse column pass), where it differs: column length 512 runs the
# four-column kernels (k_inv_cols_symx: seven workgroup barriers per transform), 1024 and 2048 the wave-per-column
# kernels - per padded cell 512 costs the column pass ~2.5 x what 1024 does (profiles/r05_small_dems.txt)
TILE_PENALTY_Y = dict(TILE_PENALTY)


def choose_tile(n_core, span, n_global, whole_axis, t_max=T_MAX, penalty=None):
    """Pick the FFT length for one axis.

    n_core     -- output cells this device owns along the axis
    span       -- pmax - pmin of the batch (support extent minus one)
    n_global   -- DEM size along the axis
    whole_axis -- True when the device holds the whole periodic axis

    Returns (T, V, ntiles, circular).  Overlap-save: a length-T circular
    convolution yields V = T - span valid outputs.  When the axis itself is a
    power of two (and we own all of it) the DEM's own periodicity is used and
    nothing is wasted."""
    best = None

    penalty = TILE_PENALTY if penalty is None else penalty

    def cost(nt, t):
        return nt * t * (math.log2(t) + 4.0) * penalty.get(t, 1.0)

    if whole_axis and _is_pow2(n_global) and T_MIN <= n_global <= t_max \
            and span < n_global:
        best = (cost(1, n_global), n_global, n_global, 1, True)
    t = T_MIN
    while t <= t_max:
        if t > span:
            v = t - span
            nt = -(-n_core // v)
            c = cost(nt, t)
            if best is None or c < best[0]:
                best = (c, t, v, nt, False)
        t *= 2
    if best is None:
        raise ValueError("template support (%d cells) exceeds the largest "
                         "FFT tile (%d)" % (span + 1, t_max))
    return best[1:]


class Plan(object):
    """Geometry of one sc_match call (mirrors struct sc_plan)."""

    def __init__(self, ny, nx, core, bbox, whole=True, method=METHOD_FFT,
                 group=1, t_max=T_MAX):
        cy0, cy1, cx0, cx1 = core
        pmin, pmax, qmin, qmax = bbox
        self.ny, self.nx = ny, nx
        self.core = core
        self.bbox = bbox
        self.oy, self.ox = ny % 2, nx % 2
        self.method = method
        self.group = group
        self.Py, self.Qx = pmax, qmax
        if method == METHOD_FFT:
            self.Ty, self.Vy, self.nty, self.circ_y = choose_tile(
                cy1 - cy0, pmax - pmin, ny, whole and (cy1 - cy0) == ny, t_max, TILE_PENALTY_Y)
            self.Tx, self.Vx, self.ntx, self.circ_x = choose_tile(
                cx1 - cx0, qmax - qmin, nx, whole and (cx1 - cx0) == nx, t_max)
            # a circular axis has no halo to place: any origin with pmax <= P <= T + pmin serves.
            # T/2 serves every template that fits the grid, so that the tile - and with it the
            # curvature spectra a search keeps (sc_set_option "spectra_mb") - is the same for
            # every scale of a multi-scale job
            # ONE tile of column length 2048 (a 2048 x 2048 DEM: BASELINE config C2) is the inverse column
            # pass's worst case: no second tile to pair with, so templates ride in pairs, and at column length
            # 2048 two coefficient planes next to the parked spectrum only fit a wave that has a SIMD to itself
            # (k_inv_cols_w4: 1.9 x the time per output of the eight-wave kernel).  Three tiles of column
            # length 1024 - a pair on the eight-wave kernel and a paired-template tile that fits it at 1024 -
            # transform 1.5 x the cells and are still faster: 33.9 ms against 37.3 at C2
            # (tools/plan_lab.py, profiles/r04_c2_plans.txt).
            if self.nty * self.ntx == 1 and self.Ty == 2048 and t_max >= 2048:
                try:
                    alt = choose_tile(cy1 - cy0, pmax - pmin, ny, False, 1024)
                except ValueError:                  # (a support no 1024-tile holds)
                    alt = None
                if alt is not None and alt[0] == 1024 and alt[2] * 1024 <= 1.6 * self.Ty:
                    self.Ty, self.Vy, self.nty, self.circ_y = alt
            if self.circ_y:
                self.Py = self.Ty // 2
            if self.circ_x:
                self.Qx = self.Tx // 2
        else:
            self.Ty = self.Tx = self.Vy = self.Vx = 0
            self.nty = self.ntx = 0
            self.circ_y = self.circ_x = False

    def tiles(self):
        """[(i0, j0, vy, vx, gi0, gj0)]: core origin, valid extent and the
        global origin of the tile's input window."""
        cy0, cy1, cx0, cx1 = self.core
        out = []
        for ty in range(self.nty):
            i0 = cy0 + ty * self.Vy
            vy = min(self.Vy, cy1 - i0)
            for tx in range(self.ntx):
                j0 = cx0 + tx * self.Vx
                vx = min(self.Vx, cx1 - j0)
                out.append((i0, j0, vy, vx, i0 + self.oy - self.Py,
                            j0 + self.ox - self.Qx))
        return out

    def padded_cells(self):
        return self.nty * self.ntx * self.Ty * self.Tx

    def __repr__(self):
        return ("Plan(T=%dx%d V=%dx%d tiles=%dx%d circ=%s/%s bbox=%s)"
                % (self.Ty, self.Tx, self.Vy, self.Vx, self.nty, self.ntx,
                   self.circ_y, self.circ_x, (self.bbox,)))


# real-space kernel (k_direct2): a 256-cell wide patch plus the template window, padded to groups
# of four taps, must fit 16 rows of its 39.5 K-float LDS slab - and the box kernel kept for
# cross-checks its own slab (sc_kernels.hip direct_window_fits)
DIRECT_MAX_WINDOW = 2208


def direct_window_fits(ww):
    return (256 + ((ww + 3) & ~3)) * 16 <= 39 * 1024 + 512 and ((64 + ww - 1) | 1) * 16 <= 36 * 1024


def direct_cost(n_taps):
    """Time per output cell and template of the real-space path, picoseconds.  Round 3, k_direct2
    (device time at 4096^2, profiles/r03_crossover.txt): 0.27 ms at 46 taps, 1.34 ms at 930, 4.7 ms
    at 4652, 9.0 ms at 9304, 25.7 ms at 29 424 - 16 ps + 0.055 ps per tap for windows whose rows
    run 16 taps or more (T3 shared between adjacent outputs), up to 0.12 ps per tap for the
    thinnest (a 5-cell run is padded to two groups of four); 0.07 here.
    (Round 2's box kernel: 60 + 1.9 ps per BOX cell, 72.8 ms at 928 taps.)"""
    return 16.0 + 0.07 * n_taps


def fft_cost(plan, n_cells, n_params=1):
    """The same for the FFT path: per PADDED cell ~24 ps of work that an orientation does once
    (curvature transforms, launches) plus ~4.4 ps per template (0.56 - 0.64 ms per template at
    4096^2 with one age per orientation whatever the support; 5.7 ps per cell and template at
    10000^2 with 35 ages), times the tile-size penalty of tiles below 512.  The real-space path
    wins for single-age searches below ~250 taps, for templates no tile holds, and it is the
    exact one on surfaces without a noise floor."""
    per_template = 4.4 + 24.0 / max(1, n_params)
    return per_template * math.sqrt(TILE_PENALTY.get(plan.Ty, 1.0) * TILE_PENALTY.get(plan.Tx, 1.0)) \
        * plan.padded_cells() / float(n_cells)
