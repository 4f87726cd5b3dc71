"""ctypes binding of libscarplet_host.so (include/scarplet_host.h): host-only helpers of the
GeoTIFF reader, built with gcc - no HIP, no RCCL, so that ``sl.load`` / ``DEMGrid(filename)``
work on a box without ROCm.  Where the library has not been built the decoder below does the
same work in Python (slowly)."""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscarplet_host.so")

# every symbol include/scarplet_host.h declares: (restype, argtypes)
SIGNATURES = {
    "sch_tiff_lzw_decode": (C.c_longlong, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
}

_lib = None


def load():
    """The shared library with every function bound, or None when it is not built."""
    global _lib
    if _lib is None and os.path.exists(LIB_PATH):
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _lzw_decode_py(src, cap):
    """TIFF 6.0 section 13 in Python: the same decoder as sch_tiff_lzw_decode, same return codes."""
    table = [bytes((i,)) for i in range(256)] + [b"", b""]
    out = bytearray()
    nbits, old, bitpos, total = 9, None, 0, len(src) * 8
    while bitpos + nbits <= total:
        byte = bitpos >> 3
        w = int.from_bytes(src[byte:byte + 3].ljust(3, b"\0"), "big")
        code = (w >> (24 - nbits - (bitpos & 7))) & ((1 << nbits) - 1)
        bitpos += nbits
        if code == 257:
            break
        if code == 256:
            del table[258:]
            nbits, old = 9, None
            continue
        if old is None:
            if code > 255:
                return -1
            entry = table[code]
        elif code < len(table):
            entry = table[code]
            if len(table) < 4096:
                table.append(old + entry[:1])
        elif code == len(table) and len(table) < 4096:
            entry = old + old[:1]
            table.append(entry)
        else:
            return -1
        if len(out) + len(entry) > cap:
            return -2
        out += entry
        old = entry
        if len(table) >= (1 << nbits) - 1 and nbits < 12:
            nbits += 1
    return bytes(out)


def tiff_lzw_decode(raw, nbytes):
    """One LZW strip / tile of a TIFF (Compression = 5) -> a uint8 array of ``nbytes`` bytes."""
    nbytes = int(nbytes)
    src = np.frombuffer(raw, dtype=np.uint8)
    lib = load()
    if lib is not None:
        out = np.empty(nbytes, dtype=np.uint8)
        n = lib.sch_tiff_lzw_decode(src.ctypes.data, src.size, out.ctypes.data, out.size)
    else:
        res = _lzw_decode_py(src.tobytes(), nbytes)
        n = res if isinstance(res, int) else len(res)
        out = None if isinstance(res, int) else np.frombuffer(res, dtype=np.uint8)
    if n == -2 or n > nbytes:
        raise ValueError("LZW strip decodes to more than the %d bytes its geometry allows" % nbytes)
    if n < 0:
        raise ValueError("malformed LZW stream")
    if n < nbytes:
        raise ValueError("LZW strip decodes to %d bytes, %d expected" % (n, nbytes))
    return out
