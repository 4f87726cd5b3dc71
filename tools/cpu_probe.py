#!/usr/bin/env python3
"""How the oracle scales on this box's host cores (calibrates bench.py's cpu_baseline):
one whole-DEM template single-threaded, a process pool, and threaded FFTs."""
import multiprocessing as mp, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import scarplet_oracle as orc
from scarplet_amd import synthetic, _plan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cores = len(os.sched_getaffinity(0))
mem = [l for l in open("/proc/meminfo") if l.startswith(("MemTotal", "MemAvailable"))]
print("cores", cores, "".join(mem).replace("\n", " "))
Z = np.ascontiguousarray(synthetic.synthetic_scarp(n)._griddata, dtype=np.float64)
ages, angs = _plan.age_grid(), _plan.angle_grid()

def one(job):
    age, ang, w = job
    t0 = time.time()
    orc.match_template(Z, 1.0, 1.0, orc.SCARP, 100, age, ang, workers=w)
    return time.time() - t0

sys.path.insert(0, ROOT)
from bench import mem_available_bytes           # min(MemAvailable, cgroup limit - usage)
max_procs = int(0.6 * mem_available_bytes() // (120.0 * n * n + (64 << 20)))     # ~12 GB per 10000^2 template
print("memory allows %d whole-DEM templates at a time" % max_procs)
for procs, thr in [(1, 1), (1, 16), (8, 4), (16, 1), (32, 1)]:
    if procs * thr > cores or procs > max_procs:
        continue
    jobs = [(float(ages[(3 * k) % 35]), float(angs[(37 * k) % 181]), thr) for k in range(procs)]
    t0 = time.time()
    if procs == 1:
        per = [one(jobs[0])]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            per = pool.map(one, jobs, chunksize=1)
    dt = time.time() - t0
    print("procs %3d x threads %2d: wall %.1f s, mean per template %.1f s -> %.2f Mpx.template/s" % (
        procs, thr, dt, np.mean(per), n * n * procs / dt / 1e6), flush=True)
