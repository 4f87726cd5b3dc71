#!/usr/bin/env python3
"""Where sl.match's time outside the search goes (10000 x 10000), and what the host link gives:
raw hipMemcpy rates pageable / registered / hipHostMalloc'd, both directions."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

hip = C.CDLL("libamdhip64.so")
def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))

def raw_rates(nbytes=1 << 30):
    dev = C.c_void_p()
    chk(hip.hipMalloc(C.byref(dev), C.c_size_t(nbytes)), "hipMalloc")
    out = {}
    def rate(fn, reps=3):
        fn()
        hip.hipDeviceSynchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        hip.hipDeviceSynchronize()
        return nbytes * reps / (time.perf_counter() - t0) / 1e9
    a = np.ones(nbytes // 8)                               # pageable, touched
    pa = C.c_void_p(a.ctypes.data)
    out["pageable H2D"] = rate(lambda: hip.hipMemcpy(dev, pa, C.c_size_t(nbytes), 1))
    out["pageable D2H"] = rate(lambda: hip.hipMemcpy(pa, dev, C.c_size_t(nbytes), 2))
    t0 = time.perf_counter()
    b = np.empty(nbytes // 8)
    pb = C.c_void_p(b.ctypes.data)
    hip.hipMemcpy(pb, dev, C.c_size_t(nbytes), 2)
    out["pageable D2H into FRESH np.empty (incl. page faults)"] = nbytes / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    chk(hip.hipHostRegister(pa, C.c_size_t(nbytes), 0), "hipHostRegister")
    out["hipHostRegister GB/s of registration"] = nbytes / (time.perf_counter() - t0) / 1e9
    out["registered H2D"] = rate(lambda: hip.hipMemcpy(dev, pa, C.c_size_t(nbytes), 1))
    out["registered D2H"] = rate(lambda: hip.hipMemcpy(pa, dev, C.c_size_t(nbytes), 2))
    t0 = time.perf_counter()
    hip.hipHostUnregister(pa)
    out["hipHostUnregister GB/s"] = nbytes / (time.perf_counter() - t0) / 1e9
    ph = C.c_void_p()
    t0 = time.perf_counter()
    chk(hip.hipHostMalloc(C.byref(ph), C.c_size_t(nbytes), 0), "hipHostMalloc")
    out["hipHostMalloc GB/s of allocation"] = nbytes / (time.perf_counter() - t0) / 1e9
    out["pinned H2D"] = rate(lambda: hip.hipMemcpy(dev, ph, C.c_size_t(nbytes), 1))
    out["pinned D2H"] = rate(lambda: hip.hipMemcpy(ph, dev, C.c_size_t(nbytes), 2))
    # host-side copy out of pinned memory into a touched / a fresh pageable array
    src = (C.c_char * nbytes).from_address(ph.value)
    v = np.frombuffer(src, dtype=np.float64)
    t0 = time.perf_counter(); a[:] = v; out["memcpy pinned -> touched pageable (numpy, 1 thread)"] = nbytes / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter(); c = np.empty_like(a); c[:] = v; out["memcpy pinned -> FRESH pageable"] = nbytes / (time.perf_counter() - t0) / 1e9
    t0 = time.perf_counter()
    hip.hipHostFree(ph)
    out["hipHostFree GB/s"] = nbytes / (time.perf_counter() - t0) / 1e9
    hip.hipFree(dev)
    return out

for k, v in raw_rates().items():
    print("  %-58s %8.2f GB/s" % (k, v), flush=True)

import scarplet_amd as sl
from scarplet_amd import synthetic, _plan, core
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = synthetic.synthetic_scarp(n)
z = g._griddata
def T(label, fn):
    t0 = time.perf_counter(); r = fn(); print("  %-44s %7.1f ms" % (label, 1e3 * (time.perf_counter() - t0)), flush=True); return r
print("stages of sl.match(%d x %d):" % (n, n))
T("np.isnan(z).any()", lambda: np.isnan(z).any())
if hasattr(core, "_dem_fingerprint"):
    T("_dem_fingerprint", lambda: core._dem_fingerprint(z, 1.0, 1.0))
m = T("Matcher(g)  [first: context + upload]", lambda: sl.Matcher(g))
m.ctx.dem_key = None
T("Matcher.set_data again (upload + curvature planes)", lambda: m.set_data(g))
ages, angles = _plan.age_grid(), _plan.angle_grid()
d = T("describe 6335 templates", lambda: m.describe(sl.Scarp, 100, ages, angles))
T("search (1 orientation only, warm-up)", lambda: m.search(sl.Scarp, 100, ages, angles[:1], method="fft"))
T("search (full grid)", lambda: m.search(sl.Scarp, 100, ages, angles, method="fft"))
T("result()  [first]", lambda: m.result())
T("result()  [second]", lambda: m.result())
for rep in range(3):
    T("sl.match end to end, call %d" % rep, lambda: sl.match(g, sl.Scarp, scale=100, method="fft"))
