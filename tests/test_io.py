"""GeoTIFF reading without GDAL and the DEMGrid duck type (host side)."""
import numpy as np
import pytest

import scarplet_amd as sl
from scarplet_amd import tiff
from conftest import golden


def test_tiled_deflate_predictor_int16_geotiff():
    # crop of the reference's grandcanyon.tif, same encoding: 32x32 tiles, deflate,
    # horizontal predictor, ModelPixelScale/Tiepoint and GDAL_NODATA tags
    a, gt, nodata = tiff.read_geotiff(golden("grandcanyon_crop.tif"))
    assert a.dtype == np.int16 and np.array_equal(a, np.load(golden("grandcanyon_crop.npy")))
    assert np.isclose(gt[1], 76.43702828517416) and np.isclose(gt[5], -76.43702828516871)
    assert nodata == -32768.0


def test_stripped_float32_tiff():
    a, gt, nodata = tiff.read_geotiff(golden("carrizo_crop.tif"))
    assert a.dtype == np.float32 and np.array_equal(a, np.load(golden("carrizo_crop.npy")))
    assert gt is None and nodata is None


def test_demgrid_load_matches_reference_contract():
    g = sl.DEMGrid(golden("grandcanyon_crop.tif"))
    assert g._griddata.dtype == np.float64 and g._griddata.shape == (96, 80)      # dem.py:317
    assert np.isclose(g._georef_info.dx, 76.43702828517416)
    assert np.isclose(g._georef_info.dy, -76.43702828516871)                      # dem.py:331-332
    assert (g._georef_info.ny, g._georef_info.nx) == (96, 80)
    h = sl.DEMGrid.from_array(np.ones((5, 7), np.float32), 2.0)
    assert h._georef_info.dx == 2.0 and h._georef_info.dy == 2.0 and h.shape == (5, 7)


def test_fill_nodata_oracle_on_a_plane():
    """The restated GDALFillNodata pass (parity unpinned, see its docstring): a planar
    surface stays planar where sources surround the hole, and nothing is left NaN."""
    import scarplet_oracle as orc
    z = np.add.outer(np.arange(20.), np.arange(30.))
    truth = z.copy()
    z[4:7, 10:13] = np.nan
    z[0, 0] = np.nan
    z[12, 5:9] = np.nan
    f = orc.fill_nodata(z)
    assert not np.isnan(f).any()
    # GDAL's quadrants are not symmetric (the cell's own column feeds the left ones only, the
    # right ones start one column out): the centre of a hole in a plane is close, not exact
    assert abs(f[5, 11] - truth[5, 11]) <= 1.0
    assert np.abs(f[4:7, 10:13] - truth[4:7, 10:13]).max() <= 1.0        # inverse distance is not linear-exact
    assert abs(f[0, 0] - truth[0, 0]) <= 1.0 and np.allclose(f[12, 5:9], truth[12, 5:9], atol=0.5)
    assert np.array_equal(f[~np.isnan(z)], truth[~np.isnan(z)])          # valid cells untouched
    # a cell with no source within reach stays nodata in ONE pass
    one = orc.fill_nodata_pass(z, 0.0)
    assert np.isnan(one[5, 11])
    # the published quadrant rules on a single hole: up / down neighbours feed the LEFT quadrants
    # at step 0, the right neighbour both right quadrants at step 1, the left neighbour nothing
    w = np.array([[0., 1., 0.], [8., np.nan, 4.], [0., 2., 0.]])
    assert orc.fill_nodata_pass(w, 1.0)[1, 1] == (1. + 2. + 4. + 4.) / 4
    # float32 scanlines: a float64 grid comes back rounded through float32
    v = np.array([[0.1, 0.2, 0.3], [0.4, np.nan, 0.6], [0.7, 0.8, 0.9]])
    r = orc.fill_nodata_pass(v, 1.0)
    assert np.array_equal(r[0], v[0].astype(np.float32).astype(float)) and r[1, 1] == np.float32(r[1, 1])


def test_fill_nodata_isolated_cells_do_not_stall():
    """One bad pixel: max(nodata per row, per column) / 2 = 1 / 2, nothing lies within half a
    cell, the pass fills nothing - the reference's loop never ends there (dem.py:400).  The
    restated driver widens the search instead of giving up silently."""
    import scarplet_oracle as orc
    rng = np.random.default_rng(3)
    z = rng.standard_normal((16, 18)).astype(np.float32).astype(float)
    z[3, 4] = z[9, 12] = np.nan
    assert np.isnan(orc.fill_nodata_pass(z, 0.5)).sum() == 2
    f = orc.fill_nodata(z)
    assert not np.isnan(f).any() and np.array_equal(f[~np.isnan(z)], z[~np.isnan(z)])
    assert np.isnan(orc.fill_nodata(np.full((4, 5), np.nan))).all()       # nothing to fill from: ends


@pytest.mark.gpu
def test_fill_nodata_device_equals_oracle():
    """sc_fill_nodata against the oracle's pass, bit for bit (float32 values, float64 sums, same operation order),
    and DEMGrid._fill_nodata (the reference's repeat-until-filled loop, dem.py:388-414)."""
    import scarplet_oracle as orc
    from scarplet_amd import _lib
    rng = np.random.default_rng(12)
    z = np.cumsum(rng.standard_normal((90, 110)), 1) + 50.0
    z[rng.random(z.shape) < 0.03] = np.nan          # speckle
    z[20:33, 40:47] = np.nan                         # a hole
    z[:, 100] = np.nan                               # a dead column
    z[0:3, 0:4] = np.nan                             # a corner
    ctx = _lib.Context(0)
    for dist, smooth in ((1.0, 0), (2.5, 0), (7.0, 0), (7.0, 2)):
        mine = z.copy()
        left = ctx.fill_nodata(mine, dist, smooth)
        ref = orc.fill_nodata_pass(z, dist, smooth)
        assert np.array_equal(mine, ref, equal_nan=True), (dist, smooth, int(np.sum(mine != ref)))
        assert left == int(np.isnan(ref).sum())
    g = sl.DEMGrid.from_array(z, 1.0)
    g._fill_nodata()
    assert g.is_interpolated and not np.isnan(g._griddata).any()
    assert np.array_equal(g._griddata, orc.fill_nodata(z))
    assert np.array_equal(g.nodata_mask, np.isnan(z))
    # isolated nodata cells (search distance 1 / 2: the first pass fills nothing) are filled by the
    # widened search, exactly as the oracle's driver does it; an all-nodata grid warns and is
    # not marked interpolated
    w = np.cumsum(rng.standard_normal((24, 31)), 0)
    w[5, 6] = w[17, 20] = np.nan
    g = sl.DEMGrid.from_array(w, 1.0)
    g._fill_nodata()
    assert g.is_interpolated and np.array_equal(g._griddata, orc.fill_nodata(w))
    h = sl.DEMGrid.from_array(np.full((6, 7), np.nan), 1.0)
    with pytest.warns(UserWarning):
        h._fill_nodata()
    assert not h.is_interpolated and np.isnan(h._griddata).all()


def test_save_round_trip_keeps_grid_georeferencing_and_nodata(tmp_path):
    """DEMGrid.save (dem.py:291-306): float32 GeoTIFF that loads back to the
    same grid, geotransform and projection keys; NaN cells survive as nodata."""
    g = sl.DEMGrid(golden("grandcanyon_crop.tif"))
    g._griddata[3, 5] = np.nan
    out = str(tmp_path / "out.tif")
    g.save(out)
    h = sl.DEMGrid(out)
    assert h._griddata.shape == g._griddata.shape
    assert np.isnan(h._griddata[3, 5]) and np.isnan(h._griddata).sum() == 1
    ok = ~np.isnan(g._griddata)
    assert (h._griddata[ok] == g._griddata[ok].astype(np.float32)).all()
    assert np.allclose(h._georef_info.geo_transform, g._georef_info.geo_transform)
    assert h._georef_info.projection == g._georef_info.projection
    assert (h._georef_info.dx, h._georef_info.dy) == (g._georef_info.dx, g._georef_info.dy)


def test_write_geotiff_dtypes_and_rotated_transform(tmp_path):
    rng = np.random.default_rng(3)
    for dt in ("u1", "i2", "u4", "f4", "f8"):
        a = (rng.random((7, 11)) * 100).astype(dt)
        out = str(tmp_path / ("a_%s.tif" % dt))
        gt = (10.0, 2.0, 0.5, 20.0, -0.5, 3.0)
        tiff.write_geotiff(out, a, gt, nodata=-1.0)
        b, gt2, nd = tiff.read_geotiff(out)
        assert b.dtype == a.dtype and (a == b).all()
        assert np.allclose(gt, gt2) and nd == -1.0


def test_four_band_result_raster_round_trip(tmp_path):
    """4-band result rasters (CHANGELOG.md:20-24): amplitude, age, orientation, SNR as
    band-separate float32 planes with the DEM's georeferencing."""
    g = sl.DEMGrid(golden("grandcanyon_crop.tif"))
    rng = np.random.default_rng(5)
    res = rng.standard_normal((4,) + g._griddata.shape)
    out = str(tmp_path / "res.tif")
    g.save_results(out, res)
    arr, gt, nodata = tiff.read_geotiff(out)
    assert arr.shape == res.shape and arr.dtype == np.float32
    assert np.array_equal(arr, res.astype(np.float32))
    assert np.allclose(gt, g._georef_info.geo_transform) and nodata is None
    with pytest.raises(ValueError):
        g.save_results(out, res[:3])


def test_predictor2_on_float_samples_is_integer_differencing(tmp_path):
    """GDAL's -co PREDICTOR=2 on Float32 / Float64 differences the raw sample
    words (libtiff horAcc32/64), not the float values: decoding with a float
    cumsum would return wrong elevations without any error."""
    rng = np.random.default_rng(8)
    for dt in ("f4", "f8", "i2", "u4"):
        a = (rng.standard_normal((9, 13)) * 1e3).astype(dt)
        out = str(tmp_path / ("p2_%s.tif" % dt))
        tiff.write_geotiff(out, a, compress=True, predictor=2)
        b, _, _ = tiff.read_geotiff(out)
        assert b.dtype == a.dtype and np.array_equal(a, b)
    # the stored words really are integer differences of the float bit patterns
    a = np.array([[1.5, 2.25, -3.0]], dtype="f4")
    out = str(tmp_path / "p2_words.tif")
    tiff.write_geotiff(out, a, predictor=2)
    raw = open(out, "rb").read()[8:8 + 12]
    u = a.view("<u4")[0]
    assert np.array_equal(np.frombuffer(raw, "<u4"), np.array([u[0], u[1] - u[0], u[2] - u[1]], dtype="<u4"))


def test_plot_results_and_hillshade():
    """Row f4: the reference's result plot (core.py:380-420) and hillshade (dem.py:433-460)."""
    import matplotlib
    matplotlib.use("Agg")
    g = sl.DEMGrid(golden("grandcanyon_crop.tif"))
    hs = sl.hillshade(g)
    assert hs.shape == g._griddata.shape and 0.0 <= hs.min() and hs.max() <= 1.0
    ls = matplotlib.colors.LightSource(azdeg=315, altdeg=45)
    assert np.array_equal(hs, ls.hillshade(g._griddata, vert_exag=1, dx=g._georef_info.dx, dy=g._georef_info.dy))
    res = np.random.default_rng(0).random((4,) + g._griddata.shape)
    fig = sl.plot_results(g, res, figsize=(3, 6))
    assert len(fig.axes) == 8                       # four maps + four colour bars
    sl.Hillshade(g).plot()
    matplotlib.pyplot.close("all")


@pytest.mark.parametrize("decoder", ["host_library", "python"])
def test_lzw_geotiffs_written_by_libtiff_decode_exactly(decoder, monkeypatch):
    """Compression = 5 (what GDAL's COMPRESS=LZW writes): fixtures encoded by libtiff through
    Pillow (oracle/gen_lzw_fixtures.py) - one strip of float32, int16 with the horizontal
    predictor, float32 in several strips - decode to the arrays they were written from, through
    libscarplet_host.so (sch_tiff_lzw_decode: plain C, no ROCm) and through the Python decoder
    that stands in where that library is not built."""
    import os
    import numpy as np
    from scarplet_amd import tiff, _hostlib
    if decoder == "python":
        monkeypatch.setattr(_hostlib, "load", lambda: None)
    else:
        assert _hostlib.load() is not None, "libscarplet_host.so not built"
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    e = np.load(os.path.join(G, "lzw_expected.npz"))
    for name, key in (("lzw_f32_strips.tif", "f32"), ("lzw_i16_pred2.tif", "i16"),
                      ("lzw_f32_multistrip.tif", "multistrip"),
                      ("deflate_f32_pred3.tif", "p3"), ("lzw_f32_pred3.tif", "p3"),    # floating-point predictor
                      ("bigtiff_f32_lzw.tif", "p3")):                                  # BigTIFF (64-bit offsets)
        a, gt, nodata, geokeys = tiff.read_geotiff_full(os.path.join(G, name))
        want = e[key]
        assert a.shape == want.shape and a.dtype.itemsize == want.dtype.itemsize
        assert np.array_equal(a.astype(want.dtype) if a.dtype.kind == "f" else a.view(want.dtype), want), name
    # a truncated or corrupt stream is an error, not garbage
    with pytest.raises(ValueError):
        _hostlib.tiff_lzw_decode(b"\x80\x00", 16)                  # ClearCode, then nothing
    with pytest.raises(ValueError):
        _hostlib.tiff_lzw_decode(b"\xff\xff\xff\xff", 16)          # a code far beyond the table


def test_reading_a_dem_does_not_load_the_gpu_library():
    """sl.DEMGrid(filename) on an LZW GeoTIFF must work without the HIP / RCCL runtimes: the
    decoder lives in libscarplet_host.so, and nothing on the way opens libscarplet_hip.so."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import scarplet_amd as sl, scarplet_amd._lib as L; "
            "g = sl.DEMGrid(%r); assert g._griddata.size > 0; assert L._lib is None, 'GPU library was loaded'"
            % os.path.join(root, "tests", "golden", "lzw_f32_strips.tif"))
    subprocess.check_call([sys.executable, "-c", code], cwd=root)
