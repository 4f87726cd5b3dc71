#!/usr/bin/env python3
"""LZW GeoTIFF fixtures for tests/test_io.py, written by Pillow (= libtiff's LZW encoder, the
one GDAL uses): a float32 one-strip image, an int16 image with the horizontal predictor, and a
float32 image of several strips, all cut from the reference's carrizo sample as committed in
tests/golden/dem_carrizo.npz.  The expected arrays are stored next to them.
Run once where Pillow is available: python oracle/gen_lzw_fixtures.py"""
import os
import numpy as np
from PIL import Image, features

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
assert features.check("libtiff"), "Pillow without libtiff cannot write LZW"
z = np.load(os.path.join(G, "dem_carrizo.npz"))["z"].astype(np.float32)
f32 = np.ascontiguousarray(z[100:197, 200:331])                     # 97 x 131, several strips
i16 = np.round((z[300:420, 50:260] - z.min()) * 20).astype(np.int16)  # 120 x 210
Image.fromarray(f32).save(os.path.join(G, "lzw_f32_strips.tif"), compression="tiff_lzw")
Image.fromarray(i16.view(np.uint16)).save(os.path.join(G, "lzw_i16_pred2.tif"), compression="tiff_lzw",
                                       tiffinfo={317: 2})
big = np.ascontiguousarray(z[:230, :250])                            # libtiff cuts it into 8-KB strips
Image.fromarray(big).save(os.path.join(G, "lzw_f32_multistrip.tif"), compression="tiff_lzw")
p3 = np.ascontiguousarray(z[150:230, 300:417])                       # floating-point predictor (GDAL PREDICTOR=3)
Image.fromarray(p3).save(os.path.join(G, "deflate_f32_pred3.tif"), compression="tiff_adobe_deflate",
                         tiffinfo={317: 3})
Image.fromarray(p3).save(os.path.join(G, "lzw_f32_pred3.tif"), compression="tiff_lzw", tiffinfo={317: 3})
Image.fromarray(p3).save(os.path.join(G, "bigtiff_f32_lzw.tif"), big_tiff=True, compression="tiff_lzw")
np.savez_compressed(os.path.join(G, "lzw_expected.npz"), f32=f32, i16=i16, multistrip=big, p3=p3)
for n in ("lzw_f32_strips.tif", "lzw_i16_pred2.tif", "lzw_f32_multistrip.tif", "deflate_f32_pred3.tif",
          "lzw_f32_pred3.tif", "bigtiff_f32_lzw.tif"):
    print(n, os.path.getsize(os.path.join(G, n)))
