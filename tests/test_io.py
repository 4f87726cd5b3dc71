"""GeoTIFF reading without GDAL and the DEMGrid duck type (host side)."""
import numpy as np

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


def test_fill_nodata_leaves_no_nan():
    z = np.add.outer(np.arange(20.), np.arange(30.))
    z[4:7, 10:13] = np.nan
    z[0, 0] = np.nan
    g = sl.DEMGrid.from_array(z, 1.0)
    g._fill_nodata()
    assert not np.isnan(g._griddata).any() and g.is_interpolated
    assert abs(g._griddata[5, 11] - 16.0) < 1.0          # planar surface: the fill stays on it
