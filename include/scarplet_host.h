/*
 * scarplet_host.h - C ABI of libscarplet_host.so: host-only helpers of the GeoTIFF reader
 * (scarplet_amd/tiff.py).  Plain C, no HIP / RCCL dependency: reading a DEM works on a box
 * without ROCm.  (Until round 3 this entry point lived in libscarplet_hip.so.)
 */
#ifndef SCARPLET_HOST_H
#define SCARPLET_HOST_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* TIFF LZW (Compression = 5, what GDAL's COMPRESS=LZW writes; TIFF 6.0 section 13: MSB-first
 * codes of 9 .. 12 bits, ClearCode 256, EndOfInformation 257, the code width grows one code
 * early).  Decodes one strip / tile src[0 .. n) into dst[0 .. cap).  Returns the number of
 * bytes written, -1 for a malformed stream, -2 when dst is too small.  Replaces what the
 * reference gets from GDAL's reader (scarplet/dem.py:308-348, gdal.Open / ReadAsArray). */
long long sch_tiff_lzw_decode(const unsigned char* src, size_t n, unsigned char* dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
