"""numpy model of the device pipeline (test infrastructure).

Mirrors, step for step and index for index, what the HIP kernels in
scarplet_amd/csrc do - window synthesis from a template descriptor, tile
inputs, two-tiles-per-complex-FFT packing, the (W + iM) spectrum split, the
epilogue and the fold - but in numpy, so that the host-side planning
(scarplet_amd/_plan.py, WindowedTemplate._device_descriptor) can be checked
against the oracle on a machine without a GPU.
"""

import numpy as np

from scarplet_amd import _plan
from scarplet_amd import WindowedTemplate as WT

EPS = float(np.spacing(1))


def curvature_planes(z, dx, dy):
    z = np.asarray(z, dtype=float)
    ny, nx = z.shape
    A = np.zeros((ny, nx))
    B = np.zeros((ny, nx))
    C = np.zeros((ny, nx))
    A[:, 1:-1] = ((z[:, 2:] - z[:, 1:-1]) - (z[:, 1:-1] - z[:, :-2])) / dx ** 2
    B[1:, 1:] = ((z[1:, 1:] - z[1:, :-1]) / dx - (z[:-1, 1:] - z[:-1, :-1]) / dx) / dx
    C[1:-1, :] = ((z[2:, :] - z[1:-1, :]) - (z[1:-1, :] - z[:-2, :])) / dy ** 2
    return A, B, C


def synth_window(desc, xaxis, yaxis):
    """Device kernel k_template_windows: dense window over the support bbox.
    Returns (w, m, n, ts) with w, m of shape (pmax-pmin+1, qmax-qmin+1)."""
    pmin, pmax, qmin, qmax = desc["bbox"]
    ny, nx = len(yaxis), len(xaxis)
    k = ny // 2 + np.arange(pmin, pmax + 1)
    l = nx // 2 + np.arange(qmin, qmax + 1)
    x = xaxis[l][np.newaxis, :]
    y = yaxis[k][:, np.newaxis]
    ca, sa = desc["cos_a"], desc["sin_a"]
    xr = x * ca + y * sa
    yr = (-x) * sa + y * ca
    inside = (np.abs(xr) < desc["c"]) & (np.abs(yr) < desc["d"])
    if desc["kind"] == WT.KIND_SCARP:
        w = (-xr / desc["p0"]) * np.exp(-(xr * xr) / desc["p1"])
        m = inside & (xr != 0)
    elif desc["kind"] == WT.KIND_RICKER:
        u2 = (desc["p0"] * xr) ** 2
        w = (1. - 2. * u2) * np.exp(-u2)
        m = inside & (u2 < WT.EXP_UNDERFLOW) & ((1. - 2. * u2) != 0)
    else:
        raise ValueError(desc["kind"])
    if desc["flags"] & WT.FLAG_NEGATE:
        w = -w
    w = np.where(m, w, 0.0)
    return w, m, float(np.sum(m)) + EPS, float(np.sum(w * w))


def epilogue(xc, t3, n, ts, desc, xaxis, yaxis, gi, gj):
    """Device epilogue for cells with global indices gi (rows, column vector)
    and gj (columns, row vector)."""
    amp = xc / ts
    T1 = ts * amp * amp
    with np.errstate(divide="ignore", invalid="ignore"):
        err = (1 / n) * (T1 - 2 * amp * xc + t3) + EPS
        snr = np.abs(T1 / err)
    f = desc["flags"]
    if f & (WT.FLAG_ERR_XR_LE0 | WT.FLAG_ERR_XR_GE0):
        xr = xaxis[gj] * desc["cos_a"] + yaxis[gi] * desc["sin_a"]
        snr = np.where((xr <= 0) if f & WT.FLAG_ERR_XR_LE0 else (xr >= 0), 0.0, snr)
    ilo, ihi, jlo, jhi = desc["limits"]
    keep = (gi >= ilo) & (gi <= ihi) & (gj >= jlo) & (gj <= jhi)
    return np.where(keep, amp, 0.0), np.where(keep, snr, 0.0)


def match_one_direct(curv, desc, xaxis, yaxis):
    """Real-space path for one template on the whole (periodic) grid."""
    ny, nx = curv.shape
    w, m, n, ts = synth_window(desc, xaxis, yaxis)
    pmin, pmax, qmin, qmax = desc["bbox"]
    oy, ox = ny % 2, nx % 2
    xc = np.zeros((ny, nx))
    t3 = np.zeros((ny, nx))
    c2 = curv * curv
    for a, p in enumerate(range(pmin, pmax + 1)):
        for b, q in enumerate(range(qmin, qmax + 1)):
            if m[a, b]:
                xc += w[a, b] * np.roll(np.roll(curv, p - oy, 0), q - ox, 1)
                t3 += np.roll(np.roll(c2, p - oy, 0), q - ox, 1)
    gi = np.arange(ny)[:, None]
    gj = np.arange(nx)[None, :]
    return epilogue(xc, t3, n, ts, desc, xaxis, yaxis, gi, gj)


def match_batch_fft(curv, descs, plan, xaxis, yaxis, cdtype=np.complex128):
    """FFT path for templates sharing one curvature plane ``curv`` (whole
    periodic grid, plan.core == whole grid).  Returns [(amp, snr)]."""
    ny, nx = curv.shape
    Ty, Tx = plan.Ty, plan.Tx
    tiles = plan.tiles()
    if len(tiles) % 2:
        tiles = tiles + [None]
    # forward transforms of tile pairs: u = tileA + i tileB (kernels F1, F2)
    rr = np.arange(Ty)[:, None]
    ss = np.arange(Tx)[None, :]
    uc, uc2 = [], []
    for a in range(0, len(tiles), 2):
        planes = []
        for t in tiles[a:a + 2]:
            if t is None:
                planes.append(np.zeros((Ty, Tx)))
                continue
            gi0, gj0 = t[4], t[5]
            planes.append(curv[(gi0 + rr) % ny, (gj0 + ss) % nx])
        u = (planes[0] + 1j * planes[1]).astype(cdtype)
        u2 = (planes[0] ** 2 + 1j * planes[1] ** 2).astype(cdtype)
        uc.append(np.fft.fft2(u).astype(cdtype))
        uc2.append(np.fft.fft2(u2).astype(cdtype))
    out = []
    fy_neg = (-np.arange(Ty)) % Ty
    fx_neg = (-np.arange(Tx)) % Tx
    for desc in descs:
        w, m, n, ts = synth_window(desc, xaxis, yaxis)
        pmin, pmax, qmin, qmax = desc["bbox"]
        # template tile: v[p % Ty, q % Tx] = W + iM (kernel F1t)
        v = np.zeros((Ty, Tx), dtype=cdtype)
        pr = np.arange(pmin, pmax + 1) % Ty
        qr = np.arange(qmin, qmax + 1) % Tx
        v[np.ix_(pr, qr)] = w + 1j * m
        vh = np.fft.fft2(v).astype(cdtype)
        vneg = np.conj(vh[np.ix_(fy_neg, fx_neg)])
        wh = 0.5 * (vh + vneg)              # FFT(W)
        mh = -0.5j * (vh - vneg)            # FFT(M)
        amp = np.zeros((ny, nx))
        snr = np.zeros((ny, nx))
        for a in range(0, len(tiles), 2):
            zw = np.fft.ifft2(wh * uc[a // 2])          # kernels I1, I2
            zm = np.fft.ifft2(mh * uc2[a // 2])
            for part, t in enumerate(tiles[a:a + 2]):
                if t is None:
                    continue
                i0, j0, vy, vx = t[:4]
                xc = (zw.imag if part else zw.real)
                t3 = (zm.imag if part else zm.real)
                ri = rr - plan.Py
                cj = ss - plan.Qx
                if plan.circ_y:
                    ri = ri % Ty
                if plan.circ_x:
                    cj = cj % Tx
                ok = (ri >= 0) & (ri < vy) & (cj >= 0) & (cj < vx)
                gi = np.clip(i0 + ri, 0, ny - 1)
                gj = np.clip(j0 + cj, 0, nx - 1)
                a_t, s_t = epilogue(xc, t3, n, ts, desc, xaxis, yaxis, gi, gj)
                gi_b = np.broadcast_to(gi, ok.shape)
                gj_b = np.broadcast_to(gj, ok.shape)
                amp[gi_b[ok], gj_b[ok]] = a_t[ok]
                snr[gi_b[ok], gj_b[ok]] = s_t[ok]
        out.append((amp, snr))
    return out
