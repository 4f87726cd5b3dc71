"""GeoTIFF reading without GDAL and the DEMGrid duck type (host side)."""
import os

import numpy as np
import pytest

import scarplet_amd as sl
from scarplet_amd import tiff
from conftest import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


# ---- untrusted input: the sanitizer leg -----------------------------------
LZW_FIXTURES = ["lzw_i16_pred2.tif", "lzw_f32_pred3.tif", "lzw_f32_strips.tif", "bigtiff_f32_lzw.tif",
                "lzw_f32_multistrip.tif"]


def _raw_strips(path):
    """The compressed strips / tiles of a TIFF fixture, byte for byte as stored."""
    import struct
    buf = open(path, "rb").read()
    bo = "<" if buf[:2] == b"II" else ">"
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 43:
        (ifd,) = struct.unpack_from(bo + "Q", buf, 8)
        t = tiff._read_ifd(buf, ifd, bo, big=True)
    else:
        (ifd,) = struct.unpack_from(bo + "I", buf, 4)
        t = tiff._read_ifd(buf, ifd, bo)
    offs, cnts = (t[324], t[325]) if 322 in t else (t[273], t[279])
    return [buf[o:o + c] for o, c in zip(offs, cnts)]


def test_lzw_decoder_under_address_sanitizer(tmp_path):
    """`make asan`: sc_host.c built with -fsanitize=address,undefined (CPU only) and driven by
    tools/lzw_fuzz.c - 10 000 mutated strips of the LZW fixtures decoded into heap blocks of exactly
    the capacity handed to the decoder; any byte read or written out of bounds, any undefined
    shift or overflow aborts the run."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    csrc = os.path.join(ROOT, "scarplet_amd", "csrc")
    subprocess.check_call(["make", "-s", "-C", csrc, "asan"])
    files = []
    for name in LZW_FIXTURES:
        for k, raw in enumerate(_raw_strips(golden(name))[:6]):
            f = tmp_path / ("%s.%d.lzw" % (name, k))
            f.write_bytes(raw)
            files.append(str(f))
    assert len(files) >= 6
    exe = os.path.join(ROOT, "tools", "bin", "lzw_fuzz_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, "10000", "20261004"] + files, capture_output=True, text=True, env=env, timeout=600)
    print(r.stdout.strip())
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "no sanitizer report" in r.stdout and "ERROR" not in r.stderr


def test_geotiff_reader_answers_corrupt_files_with_valueerror(tmp_path):
    """The Python reader over mutated copies of the fixtures (3 000 files: flipped bits, random
    bytes in the header and directory, truncations): an array or a ValueError - no other exception
    type, no allocation sized by a corrupt field, both LZW decoders in agreement."""
    rng = np.random.default_rng(99)
    seeds = [open(golden(n), "rb").read() for n in
             ("lzw_i16_pred2.tif", "carrizo_crop.tif", "grandcanyon_crop.tif", "deflate_f32_pred3.tif",
              "lzw_f32_pred3.tif", "bigtiff_f32_lzw.tif")]
    path = str(tmp_path / "m.tif")
    n_ok = n_err = 0
    for case in range(3000):
        b = bytearray(seeds[case % len(seeds)])
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 4))
            # the header and the directory (the first and last few hundred bytes) are hit most
            if rng.random() < 0.6:
                at = int(rng.integers(0, min(len(b), 600))) if rng.random() < 0.5 else len(b) - 1 - int(rng.integers(0, min(len(b), 600)))
            else:
                at = int(rng.integers(0, len(b)))
            if kind == 0:
                b[at] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b[at] = int(rng.integers(0, 256))
            elif kind == 2:
                b[at:at + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
            elif len(b) > 64:
                del b[max(32, at):]
        open(path, "wb").write(bytes(b))
        try:
            a, gt, nodata = tiff.read_geotiff(path)
            assert a.ndim in (2, 3) and a.size <= 1 << 26
            n_ok += 1
        except ValueError:
            n_err += 1
    print("corrupt GeoTIFFs: %d read, %d refused with ValueError" % (n_ok, n_err))
    assert n_ok + n_err == 3000 and n_err > 100


def test_lzw_decoders_agree_on_mutated_strips():
    """The C decoder and its Python stand-in return the same bytes or the same error code on
    600 mutated strips."""
    from scarplet_amd import _hostlib
    lib = _hostlib.load()
    if lib is None:
        pytest.skip("libscarplet_host.so not built")
    rng = np.random.default_rng(5)
    strips = [s for n in LZW_FIXTURES[:3] for s in _raw_strips(golden(n))[:2]]
    for case in range(600):
        b = bytearray(strips[case % len(strips)][:4000])
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, len(b)))
            b[at] = int(rng.integers(0, 256)) if rng.random() < 0.7 else b[at] ^ (1 << int(rng.integers(0, 8)))
        cap = int(rng.integers(0, 40000))
        src = np.frombuffer(bytes(b), dtype=np.uint8)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = lib.sch_tiff_lzw_decode(src.ctypes.data, src.size, out.ctypes.data, cap)
        ref = _hostlib._lzw_decode_py(bytes(b), cap)
        if isinstance(ref, int):
            assert n == ref, (case, n, ref)
        else:
            assert n == len(ref) and out[:n].tobytes() == ref, (case, n, len(ref))
