"""Elevation grids for the matcher (host side).

Drop-in for the parts of the reference's ``scarplet/dem.py`` that the
template-matching path touches: a ``DEMGrid`` carries ``_griddata`` (2-D
float64 elevations) and ``_georef_info.dx/.dy`` (dem.py:203-218, 317,
331-332).  Any object with those attributes - including the reference's own
``DEMGrid`` - can be handed to ``scarplet_amd.match``.

GDAL is not needed: grids are built from arrays, ``.npy`` files or GeoTIFFs
read by the small reader in ``scarplet_amd/tiff.py``.
"""

import os

import numpy as np

FLOAT32_MIN = np.finfo(np.float32).min


class GeorefInfo(object):
    """Georeferencing record with the reference's field names
    (dem.py:203-218)."""

    def __init__(self):
        self.geo_transform = None
        self.projection = None
        self.xllcenter = None
        self.yllcenter = None
        self.dx = None
        self.dy = None
        self.nx = None
        self.ny = None
        self.ulx = None
        self.uly = None
        self.lrx = None
        self.lry = None


class DEMGrid(object):
    """Grid of elevation values.

    ``DEMGrid(filename)`` loads ``.npy`` or GeoTIFF; ``DEMGrid.from_array``
    wraps an array.  Nodata cells (``FLOAT32_MIN`` or the file's nodata tag)
    become NaN like in the reference (dem.py:319-322, 361)."""

    def __init__(self, filename=None):
        self._georef_info = GeorefInfo()
        self.filename = filename
        self.label = ''
        self.is_interpolated = False
        self.nodata_value = np.nan
        if filename is None:
            self._griddata = np.empty((0, 0))
            self.shape = (0, 0)
        else:
            self.load(filename)

    @classmethod
    def from_array(cls, z, dx=1.0, dy=None):
        g = cls()
        g._set(np.array(z, dtype=float), dx, dx if dy is None else dy)
        return g

    def _set(self, z, dx, dy, geo_transform=None):
        self._griddata = z
        self.shape = z.shape
        gi = self._georef_info
        gi.dx, gi.dy = dx, dy
        gi.ny, gi.nx = z.shape
        gi.geo_transform = geo_transform if geo_transform is not None \
            else (0.0, dx, 0.0, 0.0, 0.0, dy)
        gt = gi.geo_transform
        gi.ulx, gi.uly = gt[0], gt[3]
        gi.lrx = gt[0] + dx * gi.nx
        gi.lry = gt[3] + dy * gi.ny
        gi.xllcenter = gt[0] + dx
        gi.yllcenter = gt[3] - (gi.ny + 1) * abs(dy)

    def load(self, filename):
        """Load a grid (dem.py:308-348 without GDAL)."""
        self.label = os.path.basename(filename).split('.')[0]
        ext = os.path.splitext(filename)[1].lower()
        if ext == '.npy':
            self._set(np.load(filename).astype(float), 1.0, 1.0)
        elif ext in ('.tif', '.tiff'):
            from scarplet_amd import tiff
            z, gt, nodata, geokeys = tiff.read_geotiff_full(filename)
            self._georef_info.projection = geokeys
            z = z.astype(float)
            if nodata is not None:
                z[z == nodata] = np.nan
            dx, dy = (gt[1], gt[5]) if gt is not None else (1.0, 1.0)
            self._set(z, dx, dy, gt)
        else:
            raise ValueError("unsupported grid file: %s" % filename)
        self._griddata[self._griddata == FLOAT32_MIN] = np.nan
        self.filename = filename

    def save(self, filename):
        """Save the grid as a georeferenced TIFF (dem.py:291-306 without
        GDAL): float32 samples like the reference's GDT_Float32 grids, NaN
        cells written as the float32 nodata value the loader maps back."""
        from scarplet_amd import tiff
        z = np.where(np.isnan(self._griddata), FLOAT32_MIN,
                     self._griddata).astype(np.float32)
        gi = self._georef_info
        proj = gi.projection if isinstance(gi.projection, dict) else None
        tiff.write_geotiff(filename, z, gi.geo_transform,
                           nodata=float(FLOAT32_MIN), geokeys=proj)

    def save_results(self, filename, results):
        """Write a search result as the 4-band float32 raster the reference's
        pipeline publishes (CHANGELOG.md:20-24: 1 = amplitude, 2 = relative age,
        3 = orientation, 4 = SNR) with this grid's georeferencing."""
        from scarplet_amd import tiff
        res = np.asarray(results, dtype=np.float32)
        if res.shape != (4,) + tuple(self._griddata.shape):
            raise ValueError("results must be (4, ny, nx) for this grid")
        gi = self._georef_info
        proj = gi.projection if isinstance(gi.projection, dict) else None
        tiff.write_geotiff(filename, res, gi.geo_transform, geokeys=proj)

    # -- curvature (the data-object half of the matcher's contract, core.py:341) ------------
    def _calculate_laplacian(self, device=0):
        """Curvature of the grid in the y direction (dem.py:62-66)."""
        return self._calculate_directional_laplacian(0, device=device)

    def _calculate_directional_laplacian(self, alpha, device=0):
        """Curvature of the grid in direction ``alpha`` (dem.py:68-107), float64, computed on the
        GPU (``sc_curvature_f64``: the reference's three finite-difference stencils with zero
        borders, dx for the cross term, and dem.py:103-104's combination in numpy's evaluation
        order).  NaN cells are zeroed before the stencils and come back as NaN in the result;
        like the reference (dem.py:85-86 writes through ``self._griddata``) the grid itself
        keeps the zeros."""
        from scarplet_amd.core import _context
        from scarplet_amd import WindowedTemplate as _WT
        z = self._griddata
        if z.dtype != np.float64 or not z.flags.c_contiguous:
            z = self._griddata = np.ascontiguousarray(z, dtype=np.float64)
        nan_idx = np.isnan(z)
        z[nan_idx] = 0
        gi = self._georef_info
        dx = float(gi.dx)
        dy = float(gi.dy if gi.dy is not None else gi.dx)
        ny, nx = z.shape
        ctx = _context(device)
        ctx.set_dem(z, dx, dy, _WT.centred_axis(nx, dx), _WT.centred_axis(ny, dx))
        del2z = ctx.curvature_f64(alpha, z.shape)
        del2z[nan_idx] = np.nan
        return del2z

    _calculate_directional_laplacian_numexpr = _calculate_directional_laplacian   # dem.py:109-150

    def _fill_nodata(self, device=0, max_passes=64):
        """Fill nodata (NaN) cells by interpolation so that the matcher's NaN-free
        precondition holds (dem.py:388-414).  Like the reference: repeat
        fillnodata with max_search_distance = max(most nodata cells in a row, in
        a column) / 2 until nothing is left.  One pass is ``sc_fill_nodata`` on
        the GPU (GDALFillNodata's four-quadrant inverse-distance search as GDAL
        publishes it, float32 work values; include/scarplet_hip.h).  GDAL /
        rasterio are not available to pin the pass against: PARITY UNPINNED
        (oracle fill_nodata_pass).

        Where a pass fills nothing - an isolated nodata cell gives a distance of
        1 / 2 and nothing lies within half a cell - the reference's loop never
        ends (dem.py:400); here the next pass searches at least one cell and
        twice as far every time after that.  A grid without a single valid cell
        cannot be filled: a warning, and ``is_interpolated`` stays False.
        A DEM without nodata cells does not touch the device."""
        z = np.ascontiguousarray(self._griddata, dtype=np.float64)
        mask = np.isnan(z)
        self.nodata_mask = mask.copy()
        if mask.any():
            from scarplet_amd.core import _context
            ctx = _context(device)
            stalled = None
            for _ in range(max_passes):
                dist = max(np.sum(mask, axis=1).max(), np.sum(mask, axis=0).max()) / 2
                if stalled is not None:
                    dist = max(dist, 1.0, 2.0 * stalled)
                before = int(mask.sum())
                left = ctx.fill_nodata(z, dist)
                if left == 0:
                    break
                if left == before:                # nothing within reach of any nodata cell
                    if dist > max(z.shape):
                        break
                    stalled = dist
                else:
                    stalled = None
                mask = np.isnan(z)
        self._griddata = z
        if np.isnan(z).any():
            import warnings
            warnings.warn("_fill_nodata: %d nodata cells could not be filled (no valid cell to "
                          "interpolate from)" % int(np.isnan(z).sum()))
            return
        self.is_interpolated = True
