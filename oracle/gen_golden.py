#!/opt/conda/bin/python3.9
"""Pin the oracle against the real reference and emit the golden fixtures.

Runs ONLY in the build container (needs /root/reference plus the Anaconda
interpreter, which has numexpr + tifffile):

    /opt/conda/bin/python3.9 oracle/gen_golden.py

What it does
  1. imports the UNMODIFIED reference package from /root/reference; the three
     third-party modules that are not installed (pyfftw, osgeo, rasterio) are
     replaced by stand-in module objects whose only used entry points map to
     numpy.fft.  No reference source is copied.
  2. checks oracle/scarplet_oracle.py against the reference's functions and
     against the reference's own golden files (scarplet/tests/results/*.npy);
     aborts on any mismatch.
  3. writes small fixtures (inputs + expected outputs, data only) under
     tests/golden/.  The fixtures travel to the GPU box; the reference does not.
"""
import os
import sys
import types
import warnings

import numpy as np

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def import_reference():
    nf = _mod('pyfftw.interfaces.numpy_fft', fft2=np.fft.fft2,
              ifft2=np.fft.ifft2, fftshift=np.fft.fftshift)
    ca = _mod('pyfftw.interfaces.cache', enable=lambda: None)
    itf = _mod('pyfftw.interfaces', numpy_fft=nf, cache=ca)
    _mod('pyfftw', interfaces=itf)
    _mod('osgeo', gdal=_mod('osgeo.gdal'),
         gdalconst=_mod('osgeo.gdalconst', GDT_Float32=6))
    _mod('rasterio')
    _mod('rasterio.fill', fillnodata=lambda a, **k: a)
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, REF)
    import scarplet as sl
    from scarplet import dem, WindowedTemplate
    return sl, dem, WindowedTemplate


def ref_grid(dem, z, dx, dy=None):
    g = dem.DEMGrid()
    g._griddata = np.array(z, dtype=float)
    g._georef_info.dx = dx
    g._georef_info.dy = dx if dy is None else dy
    g._georef_info.ny, g._georef_info.nx = g._griddata.shape
    g.shape = g._griddata.shape
    return g


def close(a, b, rtol=1e-9, atol=1e-12, what=""):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    ok = np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    err = np.nanmax(np.abs(a - b)) if a.size else 0.0
    print("  %-58s %s  max|d|=%.3g" % (what, "ok " if ok else "FAIL", err))
    if not ok:
        raise SystemExit("oracle does not match the reference: " + what)


def main():
    import tifffile
    import scarplet_oracle as orc
    sl, dem, WT = import_reference()
    os.makedirs(OUT, exist_ok=True)
    RES = os.path.join(REF, "scarplet/tests/results")
    rng = np.random.default_rng(7)

    # ------------------------------------------------------------------ K2
    print("templates vs reference classes and reference goldens")
    g_scarp = np.load(os.path.join(RES, "scarp_template.npy"))
    g_chan = np.load(os.path.join(RES, "channel_template.npy"))
    close(orc.scarp_template(100, 10, 0, 100, 100, 1), g_scarp,
          what="Scarp(100,10,0,100,100,1) vs scarp_template.npy")
    close(orc.ricker_template(100, 0.1, 0, 100, 100, 1), g_chan,
          what="Channel(100,.1,0,100,100,1) vs channel_template.npy")
    np.save(os.path.join(OUT, "ref_scarp_template.npy"), g_scarp)
    np.save(os.path.join(OUT, "ref_channel_template.npy"), g_chan)

    tcases = []
    for (cls, kind, d, p, ang, nx, ny, de) in [
            (WT.Scarp, orc.SCARP, 20, 10., 0.3, 64, 48, 1.),
            (WT.Scarp, orc.SCARP, 15, 3.2, -1.1, 51, 65, 2.),
            (WT.Scarp, orc.SCARP, 10, 1., np.pi / 2, 33, 33, 1.),
            (WT.Scarp, orc.SCARP, 10, 1., 0., 33, 33, 1.),
            (WT.Scarp, orc.SCARP, 100, 501.187, 0.7, 90, 80, 2.),
            (WT.Ricker, orc.RICKER, 10, 0.1, 0.5, 64, 64, 1.),
            (WT.Channel, orc.RICKER, 5, 0.05, -0.9, 75, 60, 1.),
            (WT.RightFacingUpperBreakScarp, orc.RIGHT_UPPER, 12, 5., 0.4, 40, 44, 1.),
            (WT.LeftFacingUpperBreakScarp, orc.LEFT_UPPER, 12, 5., -0.4, 41, 40, 1.)]:
        t = cls(d, p, ang, nx, ny, de)
        W = t.template()
        lim = t.get_window_limits()
        err = t.get_err_mask() if hasattr(t, "get_err_mask") else None
        oW, olim, oerr = orc.template_arrays(kind, d, p, ang, nx, ny, de)
        tag = "%s d=%g p=%g ang=%.2f %dx%d de=%g" % (kind, d, p, ang, nx, ny, de)
        close(oW, W, rtol=1e-13, atol=0, what="W   " + tag)
        assert np.array_equal(oW != 0, W != 0), "support differs " + tag
        assert np.array_equal(olim, lim), "window limits differ " + tag
        if err is not None:
            assert np.array_equal(oerr, err), "err mask differs " + tag
        tcases.append(dict(kind=kind, d=d, p=p, ang=ang, nx=nx, ny=ny, de=de,
                           W=W, lim=lim, err=err))
    np.savez_compressed(
        os.path.join(OUT, "ref_templates.npz"),
        n=len(tcases),
        **{"%s_%d" % (k, i): (np.array(v) if v is not None else np.zeros(0))
           for i, c in enumerate(tcases) for k, v in c.items()})

    # ------------------------------------------------------------------ K1
    print("curvature vs reference method and reference goldens (faultzone)")
    fz = tifffile.imread(os.path.join(REF, "scarplet/tests/data/faultzone.tif"))
    gfz = ref_grid(dem, fz, 2.0, 2.0)
    names = {0.0: "faultzone_del2z.npy", -np.pi / 2: "faultzone_del2z_-90.npy",
             -np.pi / 4: "faultzone_del2z_-45.npy", np.pi / 4: "faultzone_del2z_45.npy",
             np.pi / 2: "faultzone_del2z_90.npy"}
    crops = {"tl": (slice(0, 72), slice(0, 80)),
             "br": (slice(fz.shape[0] - 72, fz.shape[0]), slice(fz.shape[1] - 80, fz.shape[1]))}
    cur = {}
    for a, fn in names.items():
        gold = np.load(os.path.join(RES, fn))
        mine = orc.directional_curvature(fz, 2.0, 2.0, a)
        close(mine, gold, what="directional_curvature(%.3f) vs %s" % (a, fn))
        close(mine, gfz._calculate_directional_laplacian(a), rtol=1e-13,
              atol=0, what="   ... vs reference method")
        for cn, (sy, sx) in crops.items():
            cur["gold_%s_%d" % (cn, round(np.degrees(a)))] = gold[sy, sx]
    # the crops keep the image corner, so the oracle on the crop equals the
    # golden crop except on the crop's two interior edges.
    np.savez_compressed(os.path.join(OUT, "ref_faultzone_curvature.npz"),
                        z_tl=fz[crops["tl"]], z_br=fz[crops["br"]],
                        dx=2.0, dy=2.0, angles_deg=np.array([0, -90, -45, 45, 90]),
                        **cur)

    # ------------------------------------------------------------------ a7
    print("match_template vs reference on small grids")
    mt = []
    for (cls, kind, ny, nx, de, scale, age, ang) in [
            (WT.Scarp, orc.SCARP, 64, 64, 1., 10, 10., 0.3),
            (WT.Scarp, orc.SCARP, 61, 75, 1., 8, 3.2, -1.2),
            (WT.Scarp, orc.SCARP, 80, 64, 2., 20, 31.6, np.pi / 2),
            (WT.Scarp, orc.SCARP, 65, 65, 1., 10, 1., 0.),
            (WT.Channel, orc.RICKER, 64, 72, 1., 5, 0.1, 0.8),
            (WT.Ricker, orc.RICKER, 63, 64, 1., 8, 0.2, -0.3),
            (WT.RightFacingUpperBreakScarp, orc.RIGHT_UPPER, 64, 64, 1., 10, 10., 0.2),
            (WT.LeftFacingUpperBreakScarp, orc.LEFT_UPPER, 60, 66, 1., 10, 5., -0.6),
            (WT.Scarp, "scarp_negdy", 58, 62, 2., 16, 12., 0.9)]:
        z = np.cumsum(np.cumsum(rng.standard_normal((ny, nx)), 0), 1) * 0.01 \
            + rng.standard_normal((ny, nx)) * 0.05
        z = z.astype(np.float32)
        dy = de if kind != orc.RICKER else -de      # channels notebook uses dy=-dx
        if kind == "scarp_negdy":                   # north-up GeoTIFFs load with dy < 0 (dem.py:331-332)
            kind, dy = orc.SCARP, -de
        g = ref_grid(dem, z, de, dy)
        r_amp, r_age, r_ang, r_snr = sl.match_template(g, cls, scale, age, ang)
        o_amp, _, _, o_snr, det = orc.match_template(z, de, dy, kind, scale, age,
                                                     ang, details=True)
        tag = "%s %dx%d de=%g s=%g age=%g ang=%.2f" % (kind, ny, nx, de, scale, age, ang)
        close(o_amp, r_amp, rtol=1e-9, atol=1e-13, what="amp " + tag)
        close(o_snr, r_snr, rtol=1e-7, atol=1e-10, what="snr " + tag)
        # closed form against the FFT result
        curv = orc.directional_curvature(z, de, dy, ang)
        W, _, _ = orc.template_arrays(kind, scale, age, ang, nx, ny, de)
        close(orc.xcorr_direct(curv, W), det["xcorr"], rtol=1e-9, atol=1e-12,
              what="   real-space closed form, xcorr")
        close(orc.xcorr_direct(curv ** 2, (W != 0).astype(float)), det["T3"],
              rtol=1e-9, atol=1e-12, what="   real-space closed form, T3")
        mt.append(dict(kind=kind, z=z, dx=de, dy=dy, scale=scale, age=age,
                       ang=ang, amp=r_amp, snr=r_snr, n=det["n"],
                       ts=det["template_sum"]))
    np.savez_compressed(
        os.path.join(OUT, "ref_match_template.npz"), n=len(mt),
        **{"%s_%d" % (k, i): np.array(v) for i, c in enumerate(mt)
           for k, v in c.items()})

    # --------------------------------------------------------- a8/a9/a10
    print("full searches vs reference (synthetic.tif, reference goldens)")
    syn = tifffile.imread(os.path.join(REF, "scarplet/tests/data/synthetic.tif"))
    np.save(os.path.join(OUT, "ref_synthetic_dem.npy"), syn)
    m2 = np.load(os.path.join(RES, "synthetic_match2.npy"))
    o2 = orc.match(syn, 1.0, 1.0, orc.SCARP, scale=100, age=10,
                   ang_max=np.pi / 2, ang_min=-np.pi / 2)
    for i, nm in enumerate(["amp", "age", "angle", "snr"]):
        close(o2[i], m2[i], rtol=1e-7, atol=1e-10,
              what="match(age=10) %s vs synthetic_match2.npy" % nm)
    np.savez_compressed(os.path.join(OUT, "ref_synthetic_match2.npz"), res=m2)
    m1 = np.load(os.path.join(RES, "synthetic_match1.npy"))
    np.savez_compressed(os.path.join(OUT, "ref_synthetic_match1.npz"), res=m1)
    if "--full" in sys.argv:
        o1 = orc.match(syn, 1.0, 1.0, orc.SCARP, scale=100,
                       ang_max=np.pi / 2, ang_min=-np.pi / 2)
        for i, nm in enumerate(["amp", "age", "angle", "snr"]):
            close(o1[i], m1[i], rtol=1e-5, atol=1e-8,
                  what="match(35x181) %s vs synthetic_match1.npy" % nm)

    print("small full searches vs the reference's sl.match")
    sm = []
    for (cls, kind, n_y, n_x, de, kw) in [
            (WT.Scarp, orc.SCARP, 48, 56, 1., dict(scale=8, age=3.0, ang_max=0.4, ang_min=-0.4)),
            (WT.Scarp, orc.SCARP, 45, 45, 2., dict(scale=12, ang_max=0.1, ang_min=-0.1)),
            (WT.Channel, orc.RICKER, 40, 48, 1., dict(scale=5, age=0.1, ang_max=np.pi / 2, ang_min=-np.pi / 2))]:
        z = (np.cumsum(rng.standard_normal((n_y, n_x)), 1) * 0.05).astype(np.float32)
        g = ref_grid(dem, z, de, de)
        r = sl.match(g, cls, **kw)
        if kind == orc.RICKER:
            # even template: angle and angle+pi tie to rounding noise, so the
            # reference's own answer depends on its FFT library there.
            angs = orc.angle_grid(kw["ang_min"], kw["ang_max"])
            a_st, s_st = orc.snr_stack(z, de, de, kind, kw["scale"], [kw["age"]], angs)
            chk = orc.check_fold(r, a_st[0], s_st[0], np.full(len(angs), kw["age"]), angs,
                                 tie_rtol=1e-9, amp_tol=(1e-9, 1e-12), snr_tol=(1e-7, 1e-10))
            print("  sl.match %s %dx%d near-tie aware: bad=%d strict=%d tie=%d of %d"
                  % (kind, n_y, n_x, chk["n_bad"], chk["n_strict"], chk["n_tie"], chk["n"]))
            if chk["n_bad"]:
                raise SystemExit("oracle does not match the reference (ricker fold)")
        else:
            o = orc.match(z, de, de, kind, **kw)
            for i, nm in enumerate(["amp", "age", "angle", "snr"]):
                close(o[i], r[i], rtol=1e-7, atol=1e-10,
                      what="sl.match %s %dx%d %s %s" % (kind, n_y, n_x, sorted(kw), nm))
        sm.append(dict(kind=kind, z=z, dx=de, dy=de, res=np.stack(r),
                       keys=np.array(sorted(kw)), vals=np.array([kw[k] for k in sorted(kw)])))
    np.savez_compressed(
        os.path.join(OUT, "ref_match_small.npz"), n=len(sm),
        **{"%s_%d" % (k, i): np.array(v) for i, c in enumerate(sm)
           for k, v in c.items()})

    print("fold semantics (ties, NaN) vs reference compare()")
    a = [(np.array([[1., 2.], [3., 4.]]), 5., .1, np.array([[1., 0.], [2., np.nan]])),
         (np.array([[5., 6.], [7., 8.]]), 6., .2, np.array([[1., 3.], [1., 1.]])),
         (np.array([[9., 1.], [2., 3.]]), 7., .3, np.array([[2., 3.], [0., 9.]]))]
    r = sl.compare(iter(a), 2, 2)
    o = orc.compare(iter(a), 2, 2)
    for i in range(4):
        close(o[i], r[i], what="compare() plane %d" % i)
    np.savez_compressed(os.path.join(OUT, "ref_fold.npz"),
                        amps=np.stack([x[0] for x in a]), ages=np.array([x[1] for x in a]),
                        angs=np.array([x[2] for x in a]), snrs=np.stack([x[3] for x in a]),
                        res=np.stack(r))

    # ------------------------------------------------------------------ f1
    print("Shifted templates (WindowedTemplate.py:307-431): reference captures")
    sh = []
    for (cls, name, ny, nx, de, scale, age, ang, sdx, sdy) in [
            (WT.ShiftedLeftFacingUpperBreakScarp, "shifted_left", 52, 60, 1., 8, 6., 0.3, 3, -2),
            (WT.ShiftedRightFacingUpperBreakScarp, "shifted_right", 57, 49, 1., 7, 3., -0.8, -4, 1),
            (WT.ShiftedLeftFacingUpperBreakScarp, "shifted_left", 48, 48, 2., 14, 20., 1.2, 0, 5)]:
        z = (np.cumsum(rng.standard_normal((ny, nx)), 1) * 0.05
             + rng.standard_normal((ny, nx)) * 0.02).astype(np.float32)
        t = cls(scale, age, ang, nx, ny, de, dx=sdx, dy=sdy)
        W, lim, err = t.template(), t.get_window_limits(), t.get_err_mask()
        g = ref_grid(dem, z, de, de)
        r_amp, _, _, r_snr = sl.match_template(g, cls, scale, age, ang, dx=sdx, dy=sdy)
        curv = orc.directional_curvature(z, de, de, ang)
        o_amp, o_snr = orc.match_arrays(curv, W, lim, err)
        tag = "%s %dx%d shift (%d,%d)" % (name, ny, nx, sdx, sdy)
        close(o_amp, r_amp, rtol=1e-9, atol=1e-13, what="amp " + tag)
        close(o_snr, r_snr, rtol=1e-7, atol=1e-10, what="snr " + tag)
        sh.append(dict(name=name, z=z, de=de, scale=scale, age=age, ang=ang, sdx=sdx, sdy=sdy,
                       W=W, lim=lim, err=err, amp=r_amp, snr=r_snr))
    np.savez_compressed(
        os.path.join(OUT, "ref_shifted.npz"), n=len(sh),
        **{"%s_%d" % (k, i): np.array(v) for i, c in enumerate(sh) for k, v in c.items()})

    # ------------------------------------------------------------------ a11
    print("calculate_best_fit_parameters_serial (core.py:65-136): reference capture, Shifted template")
    rng11 = np.random.default_rng(11)            # (own stream: the fixtures above keep their draws)
    ny, nx = 48, 52
    z = (np.cumsum(rng11.standard_normal((ny, nx)), 1) * 0.05
         + rng11.standard_normal((ny, nx)) * 0.02).astype(np.float32)
    kw = dict(ang_max=0.05, ang_min=-0.05, dx=2, dy=1)
    r = sl.calculate_best_fit_parameters_serial(ref_grid(dem, z, 1.0, 1.0),
                                                WT.ShiftedLeftFacingUpperBreakScarp, 6, **kw)
    angs = orc.angle_grid(kw["ang_min"], kw["ang_max"])
    ages = orc.age_grid()
    # the oracle's fold over the reference's own template objects, angle-outer / age-inner
    def serial_results():
        for ang in angs:
            curv = orc.directional_curvature(z, 1.0, 1.0, ang)
            for age in ages:
                t = WT.ShiftedLeftFacingUpperBreakScarp(6, age, ang, nx, ny, 1.0, dx=2, dy=1)
                a, s_ = orc.match_arrays(curv, t.template(), t.get_window_limits(), t.get_err_mask())
                yield a, age, ang, s_
    o = orc.compare(serial_results(), ny, nx)
    for i, nm in enumerate(["amp", "age", "angle", "snr"]):
        close(o[i], r[i], rtol=1e-7, atol=1e-10, what="serial driver %s" % nm)
    np.savez_compressed(os.path.join(OUT, "ref_serial.npz"), z=z, de=1.0, scale=6.0,
                        ang_max=kw["ang_max"], ang_min=kw["ang_min"], sdx=kw["dx"], sdy=kw["dy"],
                        res=np.stack(r))

    # ------------------------------------------------------------------ b (plugin contract)
    print("the reference's own built-in classes are recognised (scarplet_amd.WindowedTemplate.builtin_twin)")
    sys.path.insert(0, ROOT)
    from scarplet_amd import WindowedTemplate as OWT
    for nm in ("Scarp", "RightFacingUpperBreakScarp", "LeftFacingUpperBreakScarp", "Ricker", "Channel"):
        twin = OWT.builtin_twin(getattr(WT, nm))
        print("  %-58s %s" % ("reference %s -> device descriptor of" % nm, twin.__name__ if twin else "NONE"))
        if twin is not getattr(OWT, nm):
            raise SystemExit("builtin_twin does not recognise the reference's " + nm)
    for nm in ("ShiftedLeftFacingUpperBreakScarp", "Crater", "WindowedTemplate"):
        if OWT.builtin_twin(getattr(WT, nm)) is not None:
            raise SystemExit("builtin_twin takes the reference's %s for a built-in" % nm)
    g = OWT.grid_descriptors(WT.Scarp, 100, orc.age_grid(), orc.angle_grid(-0.3, 0.3), 505, 900, 2.0)
    h = OWT.grid_descriptors(OWT.Scarp, 100, orc.age_grid(), orc.angle_grid(-0.3, 0.3), 505, 900, 2.0)
    assert all(np.array_equal(np.asarray(g[k]), np.asarray(h[k])) for k in g)

    # ------------------------------------------------------------------ sample DEMs
    print("sample DEM fixtures (rasters of scarplet/datasets/data as arrays; crops as TIFF)")
    DATA = os.path.join(REF, "scarplet/datasets/data")
    car = tifffile.imread(os.path.join(DATA, "carrizo.tif"))
    gc = tifffile.imread(os.path.join(DATA, "grandcanyon.tif"))
    assert car.shape == (900, 505) and car.dtype == np.float32
    assert gc.shape == (512, 512) and gc.dtype == np.int16
    np.savez_compressed(os.path.join(OUT, "dem_carrizo.npz"), z=car, dx=2.0, dy=2.0,
                        source="scarplet/datasets/data/carrizo.tif (B4 lidar, Wallace Creek); "
                               "900x505 float32, 2 m")
    np.savez_compressed(os.path.join(OUT, "dem_grandcanyon.npz"), z=gc, dx=1.0, dy=-1.0,
                        source="scarplet/datasets/data/grandcanyon.tif (AWS terrain tiles); 512x512 "
                               "int16; dx=1, dy=-1 as in docs/source/examples/channels.ipynb")
    # crops re-encoded the way the originals are (the GeoTIFF reader's cases):
    # carrizo: uncompressed float32 strips; grandcanyon: 32x32 tiles, deflate,
    # horizontal predictor, ModelPixelScale / ModelTiepoint / GDAL_NODATA tags
    cc = np.ascontiguousarray(car[100:150, 200:260])
    tifffile.imwrite(os.path.join(OUT, "carrizo_crop.tif"), cc, rowsperstrip=4)
    np.save(os.path.join(OUT, "carrizo_crop.npy"), cc)
    with tifffile.TiffFile(os.path.join(DATA, "grandcanyon.tif")) as tf:
        tags = tf.pages[0].tags
        scale_tag = tuple(tags["ModelPixelScaleTag"].value)
        tie = tuple(tags["ModelTiepointTag"].value)
        nod = tags["GDAL_NODATA"].value
    gcc = np.ascontiguousarray(gc[200:296, 300:380])
    tifffile.imwrite(os.path.join(OUT, "grandcanyon_crop.tif"), gcc, tile=(32, 32),
                     compression="adobe_deflate", predictor=True,
                     extratags=[(33550, "d", 3, scale_tag, False), (33922, "d", 6, tie, False),
                                (42113, "s", 0, str(nod), False)])
    np.save(os.path.join(OUT, "grandcanyon_crop.npy"), gcc)

    print("search grids")
    for lo, hi in [(-np.pi / 2, np.pi / 2), (-np.pi / 4, np.pi / 4),
                   (-17 * np.pi / 180, 17 * np.pi / 180), (-0.4, 0.4)]:
        num = int((180 / np.pi) * (hi - lo) / 1 + 1)
        assert len(orc.angle_grid(lo, hi)) == num
    assert len(orc.age_grid()) == 35
    print("all reference checks passed; fixtures in", OUT)


if __name__ == "__main__":
    main()
