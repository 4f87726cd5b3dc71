"""Minimal GeoTIFF reader and writer (no GDAL).

Covers what single-band DEM rasters use in practice and what the reference's
sample datasets need (scarplet/datasets/data/*.tif): classic TIFF and BigTIFF,
either byte order, strips or tiles, uncompressed, deflate or LZW, horizontal
and floating-point predictors, 8/16/32/64-bit integer or float samples, and the GeoTIFF tags
that define the geotransform (ModelPixelScale + ModelTiepoint, or
ModelTransformation) plus GDAL's nodata tag.  Replaces the read half of
BaseSpatialGrid.load (dem.py:308-348); anything outside this subset raises.
``write_geotiff`` is the counterpart of BaseSpatialGrid.save (dem.py:291-306):
one uncompressed strip image with the geotransform tags, the nodata tag and
the source file's GeoKey directory (the projection) passed through verbatim.
"""

import struct
import zlib

import numpy as np

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8),
          6: ("b", 1), 7: ("B", 1), 8: ("h", 2), 9: ("i", 4), 10: ("ii", 8),
          11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8), 18: ("Q", 8)}


def _read_ifd(buf, off, bo, big=False):
    """Tags of the image file directory at ``off``; ``big``: BigTIFF (64-bit counts and offsets,
    20-byte entries)."""
    if big:
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        head, entry, efmt, inline, pfmt = 8, 20, "HHQ8s", 8, "Q"
    else:
        (n,) = struct.unpack_from(bo + "H", buf, off)
        head, entry, efmt, inline, pfmt = 2, 12, "HHI4s", 4, "I"
    tags = {}
    for k in range(n):
        tag, typ, cnt, val = struct.unpack_from(bo + efmt, buf, off + head + entry * k)
        fmt, size = _TYPES.get(typ, (None, 0))
        if fmt is None:
            continue
        total = size * cnt
        if total > len(buf):                         # (a count the file cannot hold: before any buffer is sized by it)
            raise ValueError("malformed TIFF: tag %d claims %d bytes in a file of %d" % (tag, total, len(buf)))
        if total <= inline:
            data = val[:total]
        else:
            (ptr,) = struct.unpack(bo + pfmt, val)
            data = buf[ptr:ptr + total]
            if len(data) != total:
                raise ValueError("malformed TIFF: tag %d points beyond the end of the file" % tag)
        if typ == 2:
            tags[tag] = data.split(b"\0")[0].decode("ascii", "replace")
        elif typ in (5, 10):
            v = struct.unpack(bo + fmt[0] * (2 * cnt), data)
            tags[tag] = tuple(v[2 * i] / v[2 * i + 1] if v[2 * i + 1] else 0.0 for i in range(cnt))
        else:
            tags[tag] = struct.unpack(bo + fmt * cnt, data)
    return tags


GEOKEY_TAGS = (34735, 34736, 34737)      # GeoKeyDirectory, DoubleParams, AsciiParams


def read_geotiff(path):
    """Returns (array, geo_transform or None, nodata or None).

    geo_transform follows GDAL: (x0, dx, 0, y0, 0, dy) with dy negative for
    north-up rasters (dem.py:324-332 reads dx = gt[1], dy = gt[5])."""
    return read_geotiff_full(path)[:3]


# the largest decoded size a file of n bytes is allowed to claim: deflate cannot expand beyond
# 1032 : 1 and TIFF LZW (12-bit codes, strings of at most 4094 bytes) not beyond ~2730 : 1
_MAX_EXPANSION = 4096


def read_geotiff_full(path):
    """read_geotiff plus the projection: a dict {tag: value} of the GeoKey
    tags present in the file (opaque; write_geotiff stores it back).

    A file is untrusted input: whatever is wrong with it - a truncated directory, offsets
    beyond the end, a geometry its bytes cannot hold, a corrupt compressed strip - comes out as
    ValueError, never as another exception type, an allocation sized by the file's own claims
    or an endless loop (tests/test_io.py fuzzes the reader and, in an AddressSanitizer build,
    the C LZW decoder)."""
    import zlib as _z
    with open(path, "rb") as f:
        buf = f.read()
    try:
        return _read_geotiff_buffer(buf, path)
    except ValueError:
        raise
    except (struct.error, KeyError, IndexError, TypeError, OverflowError, ZeroDivisionError,
            MemoryError, _z.error) as e:
        raise ValueError("%s: malformed TIFF (%s: %s)" % (path, type(e).__name__, e))


def _read_geotiff_buffer(buf, path):
    if buf[:2] == b"II":
        bo = "<"
    elif buf[:2] == b"MM":
        bo = ">"
    else:
        raise ValueError("%s: not a TIFF file" % path)
    (magic,) = struct.unpack_from(bo + "H", buf, 2)
    if magic == 42:
        (ifd,) = struct.unpack_from(bo + "I", buf, 4)
        t = _read_ifd(buf, ifd, bo)
    elif magic == 43:                               # BigTIFF: offset size 8, then the first IFD's offset
        osize, zero, ifd = struct.unpack_from(bo + "HHQ", buf, 4)
        if osize != 8 or zero != 0:
            raise ValueError("%s: malformed BigTIFF header" % path)
        t = _read_ifd(buf, ifd, bo, big=True)
    else:
        raise ValueError("%s: unknown TIFF variant (magic %d)" % (path, magic))
    width, height = t[256][0], t[257][0]
    bits = t.get(258, (1,))[0]
    comp = t.get(259, (1,))[0]
    spp = t.get(277, (1,))[0]
    fmt = t.get(339, (1,))[0]
    pred = t.get(317, (1,))[0]
    planar = t.get(284, (1,))[0]
    if spp != 1 and planar != 2:
        raise ValueError("%s: %d interleaved samples per pixel (single-band or band-separate "
                         "rasters only)" % (path, spp))
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits not in (8, 16, 32, 64):
        raise ValueError("%s: unsupported sample format %d / %d bits" % (path, fmt, bits))
    dtype = np.dtype(bo + kind + str(bits // 8))
    if not (0 < width <= 1 << 24 and 0 < height <= 1 << 24 and 0 < spp <= 64):
        raise ValueError("%s: implausible geometry %d x %d x %d" % (path, width, height, spp))
    if width * height * spp * dtype.itemsize > max(len(buf), 1 << 16) * _MAX_EXPANSION:
        raise ValueError("%s: %d x %d x %d samples cannot come out of %d bytes"
                         % (path, width, height, spp, len(buf)))

    def decode(raw, rows, cols):
        if rows <= 0 or cols <= 0 or rows * cols * dtype.itemsize > max(len(buf), 1 << 16) * _MAX_EXPANSION:
            raise ValueError("%s: implausible strip / tile geometry %d x %d" % (path, rows, cols))
        if comp in (8, 32946):
            d = zlib.decompressobj()
            raw = d.decompress(raw, rows * cols * dtype.itemsize + 1)   # bounded: a bomb stops here
        if comp in (1, 8, 32946) and len(raw) < rows * cols * dtype.itemsize:
            raise ValueError("%s: a strip / tile holds %d bytes, %d needed"
                             % (path, len(raw), rows * cols * dtype.itemsize))
        if comp in (8, 32946):
            pass
        elif comp == 5:                             # LZW (GDAL COMPRESS=LZW)
            from scarplet_amd import _hostlib
            raw = _hostlib.tiff_lzw_decode(raw, rows * cols * dtype.itemsize)
        elif comp != 1:
            raise ValueError("%s: compression %d is not supported" % (path, comp))
        if pred == 3:
            # floating-point predictor (TIFF Technical Note 3, GDAL PREDICTOR=3): every row is
            # stored as its bytes regrouped into planes - most significant byte of every
            # sample first - and then differenced byte by byte along the whole row
            if kind != "f":
                raise ValueError("%s: predictor 3 on non-float samples" % path)
            nb = dtype.itemsize
            r = np.frombuffer(raw, dtype=np.uint8, count=rows * cols * nb).reshape(rows, cols * nb)
            r = np.cumsum(r, axis=1, dtype=np.uint8).reshape(rows, nb, cols)
            be = np.ascontiguousarray(r.transpose(0, 2, 1))          # (rows, cols, nb), big-endian bytes
            return be.view(np.dtype(">" + kind + str(nb))).reshape(rows, cols)
        a = np.frombuffer(raw, dtype=dtype, count=rows * cols).reshape(rows, cols)
        if pred == 2:
            # horizontal differencing is defined on the sample WORDS (libtiff
            # / GDAL PREDICTOR=2 on Float32 differences the raw integers):
            # accumulate as unsigned integers with wraparound, then view back
            u = np.dtype(bo + "u" + str(bits // 8))
            a = np.cumsum(a.view(u), axis=1, dtype=u).view(dtype)
        elif pred != 1:
            raise ValueError("%s: predictor %d is not supported" % (path, pred))
        return a

    if spp != 1:                                    # band-separate strips: PlanarConfiguration 2
        if 322 in t:
            raise ValueError("%s: tiled multi-band rasters are not supported" % path)
        rps = min(t.get(278, (height,))[0], height)
        if rps <= 0:
            raise ValueError("%s: %d rows per strip" % (path, rps))
        per_band = (height + rps - 1) // rps
        offs, cnts = t[273], t[279]
        if len(offs) != len(cnts) or len(offs) != per_band * spp:
            raise ValueError("%s: %d strips for %d bands of %d strips" % (path, len(offs), spp, per_band))
        out = np.empty((spp, height, width), dtype=dtype.newbyteorder("="))
        for k, (o, c) in enumerate(zip(offs, cnts)):
            b, ks = divmod(k, per_band)
            y0 = ks * rps
            h = min(rps, height - y0)
            out[b, y0:y0 + h] = decode(buf[o:o + c], h, width)
    else:
        out = np.empty((height, width), dtype=dtype.newbyteorder("="))
    if spp != 1:
        pass
    elif 322 in t:                                  # tiled
        tw, tl = t[322][0], t[323][0]
        offs, cnts = t[324], t[325]
        if tw <= 0 or tl <= 0:
            raise ValueError("%s: tile size %d x %d" % (path, tw, tl))
        across = (width + tw - 1) // tw
        if len(offs) != len(cnts) or len(offs) != across * ((height + tl - 1) // tl):
            raise ValueError("%s: %d tiles for a %d x %d image of %d x %d tiles"
                             % (path, len(offs), width, height, tw, tl))
        for k, (o, c) in enumerate(zip(offs, cnts)):
            ty, tx = divmod(k, across)
            tile = decode(buf[o:o + c], tl, tw)
            y0, x0 = ty * tl, tx * tw
            h, w = min(tl, height - y0), min(tw, width - x0)
            out[y0:y0 + h, x0:x0 + w] = tile[:h, :w]
    else:                                           # strips
        rps = min(t.get(278, (height,))[0], height)
        offs, cnts = t[273], t[279]
        if rps <= 0 or len(offs) != len(cnts) or len(offs) != (height + rps - 1) // rps:
            raise ValueError("%s: %d strips of %d rows for %d rows" % (path, len(offs), rps, height))
        for k, (o, c) in enumerate(zip(offs, cnts)):
            y0 = k * rps
            h = min(rps, height - y0)
            out[y0:y0 + h] = decode(buf[o:o + c], h, width)

    gt = None
    if 33550 in t and 33922 in t:                   # ModelPixelScale + ModelTiepoint
        sx, sy = t[33550][0], t[33550][1]
        i, j, _, x, y, _ = t[33922][:6]
        gt = (x - i * sx, sx, 0.0, y + j * sy, 0.0, -sy)
    elif 34264 in t:                                # ModelTransformation (4x4, row major)
        m = t[34264]
        gt = (m[3], m[0], m[1], m[7], m[4], m[5])
    nodata = None
    if 42113 in t:
        try:
            nodata = float(t[42113])
        except (TypeError, ValueError):
            nodata = None
        if nodata is not None and np.isnan(nodata):
            nodata = None                           # NaN cells are already NaN
    geokeys = dict((k, t[k]) for k in GEOKEY_TAGS if k in t)
    return out, gt, nodata, geokeys


def write_geotiff(path, array, geo_transform=None, nodata=None, geokeys=None,
                  compress=False, predictor=1):
    """Write a little-endian classic TIFF, one strip per band; uncompressed by
    default, ``compress=True`` deflates the strips, ``predictor=2`` stores
    horizontal differences of the sample words (integer wraparound, also for
    float samples - what libtiff / GDAL's PREDICTOR=2 do).

    array: (rows, cols) or (bands, rows, cols) - bands are stored as separate
    planes, one strip each; any of the dtypes read_geotiff accepts (float64
    grids are stored as given; pass ``array.astype('f4')`` for compact output).
    geo_transform: GDAL 6-tuple; axis-aligned transforms become
    ModelPixelScale + ModelTiepoint, rotated ones ModelTransformation."""
    a = np.ascontiguousarray(array)
    if a.ndim not in (2, 3):
        raise ValueError("write_geotiff: (rows, cols) or (bands, rows, cols) arrays only")
    bands = 1 if a.ndim == 2 else a.shape[0]
    kind = {"u": 1, "i": 2, "f": 3}.get(a.dtype.kind)
    if kind is None or a.dtype.itemsize not in (1, 2, 4, 8) or \
            (kind == 3 and a.dtype.itemsize < 4):
        raise ValueError("write_geotiff: unsupported dtype %s" % a.dtype)
    a = a.astype(a.dtype.newbyteorder("<"), copy=False)
    h, w = a.shape[-2:]
    if predictor not in (1, 2):
        raise ValueError("write_geotiff: predictor 1 or 2")
    strips = []
    for b in range(bands):
        plane = a if a.ndim == 2 else a[b]
        if predictor == 2:
            u = plane.view(np.dtype("<u%d" % a.dtype.itemsize))
            d = u.copy()
            d[:, 1:] = u[:, 1:] - u[:, :-1]          # modulo 2**bits
            plane = d
        raw = plane.tobytes()
        strips.append(zlib.compress(raw, 6) if compress else raw)
    entries = []                                    # (tag, type, count, bytes)

    def add(tag, typ, values):
        fmt = {2: None, 3: "H", 4: "I", 12: "d"}[typ]
        if typ == 2:
            data = values.encode("ascii") + b"\0"
            cnt = len(data)
        else:
            cnt = len(values)
            data = struct.pack("<%d%s" % (cnt, fmt), *values)
        entries.append((tag, typ, cnt, data))

    add(256, 4, [w])
    add(257, 4, [h])
    add(258, 3, [a.dtype.itemsize * 8] * bands)
    add(259, 3, [8 if compress else 1])
    add(262, 3, [1])                                # BlackIsZero
    offs, o = [], 8                                 # one strip per band, data at 8
    for st in strips:
        offs.append(o)
        o += len(st) + (len(st) & 1)
    data_bytes = o - 8
    add(273, 4, offs)
    add(277, 3, [bands])
    add(278, 4, [h])
    add(279, 4, [len(st) for st in strips])
    add(284, 3, [1 if bands == 1 else 2])           # band-separate planes
    if bands > 1:
        add(338, 3, [0] * (bands - 1))              # ExtraSamples: unspecified data
    if predictor == 2:
        add(317, 3, [2])
    add(339, 3, [kind] * bands)
    if geo_transform is not None:
        x0, dx, rx, y0, ry, dy = [float(v) for v in geo_transform]
        if rx == 0.0 and ry == 0.0 and dx > 0 and dy < 0:
            add(33550, 12, [dx, -dy, 0.0])
            add(33922, 12, [0.0, 0.0, 0.0, x0, y0, 0.0])
        else:
            add(34264, 12, [dx, rx, 0.0, x0, ry, dy, 0.0, y0,
                            0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])
    for tag in GEOKEY_TAGS:
        if geokeys and tag in geokeys:
            v = geokeys[tag]
            if tag == 34737:
                add(tag, 2, v if isinstance(v, str) else "".join(v))
            else:
                add(tag, 3 if tag == 34735 else 12, list(v))
    if nodata is not None:
        add(42113, 2, repr(float(nodata)))
    entries.sort(key=lambda e: e[0])

    if data_bytes + 4096 + sum(len(e[3]) for e in entries) >= 2 ** 32:
        raise ValueError("write_geotiff: raster too large for classic TIFF")
    ifd_off = 8 + data_bytes
    extra_off = ifd_off + 2 + 12 * len(entries) + 4
    ifd = struct.pack("<H", len(entries))
    extra = b""
    for tag, typ, cnt, data in entries:
        if len(data) <= 4:
            field = data.ljust(4, b"\0")
        else:
            field = struct.pack("<I", extra_off + len(extra))
            extra += data + (b"\0" if len(data) & 1 else b"")
        ifd += struct.pack("<HHI", tag, typ, cnt) + field
    ifd += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        for st in strips:
            f.write(st)
            if len(st) & 1:
                f.write(b"\0")
        f.write(ifd)
        f.write(extra)
