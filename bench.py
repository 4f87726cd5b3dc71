#!/usr/bin/env python3
"""Headline benchmark: Mpixel.template/s of one full template search.

    python bench.py --gpus N --steps K --warmup W

One "step" is one search of the 35-age x 181-orientation Scarp grid (scale
100) over the 10000 x 10000 synthetic DEM of BASELINE.md (config C3), DEM
already resident in HBM - in the mode sl.match delivers by default (--mode
exact: the float32 search with its near-ties listed, then sc_settle_exact -
the argmax (age, orientation) of every cell is the float64 reference's); the
float32 search alone (--mode float32) is timed beside it (`float32_mode`).  With N > 1 (launched by torch.distributed.run, one
rank per GPU) the same DEM is cut into a py x px tile grid (C4: 2 x 4), every
rank exchanges halos over RCCL and searches its tile: total work is fixed, so
scaling is "strong".  torch is used for the launcher contract only (rank
rendezvous, barrier, max-reduce of the timings); the product path is
ctypes -> libscarplet_hip.so.

Besides the contract's fields the JSON line carries
  config.mode, settle, float32_mode   the mode timed, the settle's counters of a
                 step, the other mode's short loop and its verification
  roofline       dominant kernel against the 28-B / 8 TB/s HBM roofline; its
                 `traffic` is the PMC-measured bytes per launch, printed only
                 when profiles/traffic.json was measured on this very .so
  cpu_baseline   the oracle timed on this box's host cores:
                 whole-DEM templates, process-pool fan-out like core.py:180-183
  end_to_end     sl.match() wall time (H2D of the DEM, descriptors, search,
                 result conversion and D2H) as Mpx.template/s
  verified       a window of the result checked against the oracle after the
                 timed loop (every template of the search, near-tie policy)

Other workloads: --config C1|C2|C5 (BASELINE.json configs as SURVEY.md 8d
defines them), --emulate-ranks R (the R rank blocks of the tiled search run one
after the other on this GPU: per-block times, their max and sum).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_UNIT = 28.0        # bytes per px.template
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
# profiling slot of the library -> kernel symbols as rocprofv3 lists them
KERNEL_SYMBOLS = {"k_inv_cols": "k_inv_cols_w8<T> (column length 512: k_inv_cols_h2; paired templates at 2048: k_inv_cols_w4)",
                  "k_inv_rows": "k_inv_rows_fast<T,false,false,false,false,NEAR> (NEAR = true in the exact mode)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--warmup-seconds", type=float, default=0.5,
                    help="warm-up steps go on until this much time has passed as well (the chip's clocks take "
                         "0.1 - 0.2 s of sustained work to come up from idle: a timed loop of a few milliseconds "
                         "ran at 1.0 GHz instead of 2.1); the headline's W steps are far beyond it, the line's "
                         "'warmup' is the number of steps actually run")
    ap.add_argument("--config", default="C3", choices=["C1", "C1F", "C2", "C3", "C5"])
    ap.add_argument("--size", dest="n", type=int, default=0, help="synthetic DEM size (default: the config's)")
    ap.add_argument("--ages", type=int, default=0, help="ages of the grid (default: the config's)")
    ap.add_argument("--angles", type=int, default=0, help="orientations (default: the config's)")
    ap.add_argument("--method", default="fft")
    ap.add_argument("--mode", default="exact", choices=["exact", "float32"],
                    help="exact (default): every step settles its near-ties in float64 on the device - the argmax (age, "
                         "orientation) of every cell is the float64 reference's, what sl.match delivers by default; float32: "
                         "the float32 search as it is (sl.match(..., exact=False))")
    ap.add_argument("--group", type=int, default=0, help="templates per inverse launch (0: auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the other_configs block (C1, C2, C5 timed and verified after the headline)")
    ap.add_argument("--other-configs-budget", type=float, default=150.0,
                    help="seconds the other_configs block may take in all (timing + oracle verification); configs "
                         "that would start beyond it are reported as skipped")
    ap.add_argument("--prof-stride", type=int, default=16)
    ap.add_argument("--opt", default="", help="engine options for lab runs: name=value[,name=value...] (sc_set_option)")
    ap.add_argument("--tile-penalty", default="", help="lab: planner weights by tile size, e.g. 512=1.5,1024=1.0 "
                                                       "(scarplet_amd._plan.TILE_PENALTY)")
    ap.add_argument("--shard", default="auto", choices=["auto", "orientations", "tiles"],
                    help="multi-rank sharding: 'orientations' = every rank the whole DEM and a chunk of the "
                         "orientation grid, records folded over RCCL (scarplet_amd.dist.OrientationMatcher); "
                         "'tiles' = BASELINE config C4, the DEM cut into rank cores with a halo exchange "
                         "(DistMatcher); auto: orientations (the benchmark DEM fits one GPU)")
    ap.add_argument("--partition", default="tiles", choices=["tiles", "grid"],
                    help="multi-rank cores: whole FFT tiles balanced over the ranks when that beats the "
                         "even grid (scarplet_amd.dist.tile_cores), or always the even grid")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="run the R blocks of the R-rank tiled search sequentially on this GPU")
    ap.add_argument("--ipc-legacy", default="auto", choices=["auto", "0", "env"],
                    help="HSA_ENABLE_IPC_MODE_LEGACY of the ranks this script starts itself (--gpus N without a "
                         "launcher): '0' sets it to 0 where the environment does not set it, 'env' leaves the "
                         "environment untouched, 'auto' tries '0' and, if the ranks fail, 'env'")
    ap.add_argument("--halo", default="rccl", choices=["rccl", "host", "gloo"],
                    help="halo exchange executor for --gpus > 1 (host: host arrays over gloo, for bring-up)")
    return ap.parse_args()


# ----------------------------------------------------------------------------- workloads
def workload(a):
    """(DEMGrid, Template, [scales], params, angles, label, oracle kind) of a BASELINE config
    (SURVEY.md section 8(d) definitions)."""
    import scarplet_amd as sl
    from scarplet_amd import _plan, synthetic
    ages35, ang181 = _plan.age_grid(), _plan.angle_grid()

    def pick(arr, k):
        return arr if not k or k >= len(arr) else arr[np.round(np.linspace(0, len(arr) - 1, k)).astype(int)]
    if a.config == "C3":
        n = a.n or 10000
        ages, angles = pick(ages35, a.ages), pick(ang181, a.angles)
        full = n == 10000 and len(ages) == 35 and len(angles) == 181
        label = "%s: %dx%d synthetic erf-scarp DEM (BASELINE.md s3), Scarp, scale=100, %d ages x %d orientations" % (
            "C3" if full else "reduced C3", n, n, len(ages), len(angles))
        return synthetic.synthetic_scarp(n), sl.Scarp, [100.0], ages, angles, label, "scarp"
    if a.config == "C2":
        n = a.n or 2048
        ages = ages35[np.round(np.linspace(0, 34, 10)).astype(int)]
        angles = pick(_plan.angle_grid(-np.pi / 4, np.pi / 4), a.angles)
        label = "C2: %dx%d synthetic DEM, Scarp, scale=100, %d ages x %d orientations" % (n, n, len(ages), len(angles))
        return synthetic.synthetic_scarp(n), sl.Scarp, [100.0], ages, angles, label, "scarp"
    f = np.load(os.path.join(ROOT, "tests", "golden",
                             "dem_carrizo.npz" if a.config in ("C1", "C1F") else "dem_grandcanyon.npz"))
    g = sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"]))
    if a.config == "C1F":
        # the reference's own flagship call, docs/source/examples/scarps.ipynb cell 12:
        # `res = sl.match(data, Scarp, scale=100.)` on load_carrizo() - "This can be slow on a laptop!"
        return g, sl.Scarp, [100.0], pick(ages35, a.ages), pick(ang181, a.angles), \
            "C1F: load_carrizo() 900x505 lidar DEM at 2 m, Scarp, scale=100, the full 35 ages x 181 orientations " \
            "(scarps.ipynb: sl.match(data, Scarp, scale=100.))", "scarp"
    if a.config == "C1":
        lim = 17 * np.pi / 180
        angles = _plan.angle_grid(-lim, lim)
        return g, sl.Scarp, [100.0], np.array([10.0]), angles, \
            "C1: load_carrizo() 900x505 lidar DEM at 2 m, Scarp, scale=100, age=10, 35 orientations", "scarp"
    return g, sl.Channel, [5.0, 10.0, 20.0, 40.0, 80.0], np.array([0.1]), pick(ang181, a.angles), \
        "C5: load_grandcanyon() 512x512 (dx=1, dy=-1), Channel f=0.1, 5 scales x 181 orientations", "ricker"


# ----------------------------------------------------------------------------- CPU baseline
def _cpu_one(job):
    age, ang, threads = job
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scarplet_oracle as orc
    z, dx, dy, kind, scale = _CPU_CTX
    t0 = time.time()
    orc.match_template(z, dx, dy, kind, scale, age, ang, workers=threads)
    return time.time() - t0


_CPU_CTX = None       # (z float64, dx, dy, kind, scale): set BEFORE the pool is forked


def cpu_quota():
    """CPUs the container's cgroup grants (cpu.max: quota / period), or None: 16 of the 256 visible on the GPU boxes."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(path).read().split()[:2]
            if q != "max":
                return float(q) / float(p)
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def host_cores():
    """Cores this process can really use: its affinity mask, capped by the cgroup's CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    return max(1, min(n, int(q + 0.5))) if q else n


def mem_available_bytes():
    """min(MemAvailable, cgroup limit - usage): what this container may still take."""
    avail = 64 << 30
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    for lim, use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            v = open(lim).read().strip()
            if v != "max" and int(v) < (1 << 60):
                avail = min(avail, int(v) - int(open(use).read().strip()))
        except (OSError, ValueError):
            pass
    return max(avail, 0)


def make_pool(g, kind, scale):
    """The host-side process pool (oracle verification + CPU baseline), forked
    HERE - before this process makes its first HIP call: a process that has
    initialised the GPU must neither fork nor exec on the GPU boxes.  The workers
    inherit the float64 DEM copy-on-write and only ever run numpy / scipy."""
    global _CPU_CTX
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scarplet_oracle  # noqa: F401  (inherited by the workers)
    _CPU_CTX = (np.ascontiguousarray(g._griddata, dtype=np.float64), float(g._georef_info.dx),
                float(g._georef_info.dy), kind, float(scale))
    return mp.get_context("fork").Pool(max(1, min(host_cores(), 64)))


def cpu_baseline(pool, g, params, angles, max_workers=32):
    """The oracle (float64 restatement of core.py:297-377, scipy.fft/pocketfft
    because pyfftw is not in the image) on THIS box's host cores, run the way
    the reference runs a search: a process pool over templates
    (core.py:180-183), every worker one whole-DEM template with
    single-threaded FFTs.  Bounded sample: one template per worker, stratified
    over the (age, orientation) grid (small and large supports, all quadrants);
    workers = min(cores, 0.6 x the container's available RAM / 12 GB per
    10000 x 10000 template, 32).  The figure is units-per-second of the sample, i.e. the full
    search's rate by extrapolation (every template costs the same FFTs)."""
    ny, nx = g._griddata.shape
    cores = host_cores()
    per_worker = 120.0 * ny * nx + (64 << 20)         # ~11 GB transient at 10000^2
    avail = mem_available_bytes()
    workers = int(max(1, min(cores, max_workers, (0.6 * avail) // per_worker)))
    n_all = len(params) * len(angles)
    n_s = max(min(8, n_all), min(2 * workers, n_all))     # two rounds of the pool
    # stratified: spread over the flattened (age-major) grid with a stride coprime to both axes
    idx = np.unique(np.round(np.linspace(0, n_all - 1, n_s)).astype(int))
    jobs = [(float(params[i // len(angles)]), float(angles[(i * 7) % len(angles)]), 1) for i in idx]
    conc = min(workers, len(jobs))
    t0 = time.time()
    per = []
    for k in range(0, len(jobs), conc):               # at most `conc` templates in flight (memory)
        per += pool.map(_cpu_one, jobs[k:k + conc], chunksize=1)
    dt = time.time() - t0
    value = ny * nx * len(jobs) / dt / 1e6
    # The memory bound leaves most cores idle on a many-core box (12 GB per whole-DEM template):
    # a second pass gives every worker scipy.fft threads for the six transforms, so that the host
    # gets its best shot (the reference itself runs single-threaded FFTs per worker)
    threaded = None
    threads = min(16, cores // max(conc, 1))
    if threads >= 2 and ny * nx >= 4_000_000:
        jobs_t = [(a_, b_, threads) for (a_, b_, _) in jobs[:conc]]
        t1 = time.time()
        pool.map(_cpu_one, jobs_t, chunksize=1)
        dt_t = time.time() - t1
        threaded = {"value": round(ny * nx * len(jobs_t) / dt_t / 1e6, 3), "unit": "Mpx.template/s",
                    "cores": int(conc * threads),
                    "sample": "%d templates at a time, %d scipy.fft threads each, wall %.1f s" % (conc, threads, dt_t)}
    return {"value": round(value, 3), "unit": "Mpx.template/s", "cores": int(conc), "kind": "port",
            "with_fft_threads": threaded,
            "sample": "%d of the %d templates (stratified over ages and orientations) on the full %dx%d DEM, "
                      "oracle/scarplet_oracle.py (float64, scipy.fft single-threaded per template), process pool: "
                      "%d templates at a time like core.py:180-183 (%d cores usable: affinity capped by the cgroup's CPU "
                      "quota; %.0f GB RAM available, 12 GB per template), wall %.1f s, mean %.1f s per template per worker; the full search's "
                      "rate is this figure by extrapolation (x%d templates)" % (
                          len(jobs), n_all, ny, nx, conc, cores, avail / 1e9, dt, float(np.mean(per)), n_all)}


# ----------------------------------------------------------------------------- verification
def verify_window(pool, res, g, kind, scale, params, angles, plan, method="fft", also=None):
    """A window of the finished search against the oracle, every template of the
    grid (oracle.snr_stack_window + check_fold, tolerances oracle.PARITY).  The
    window straddles the corner where four FFT tiles meet when the plan is tiled."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scarplet_oracle as orc
    ny, nx = g._griddata.shape
    w = 48 if min(ny, nx) >= 512 else 16
    reach = max(abs(v) for v in plan.bbox) + 3
    ci = plan.Vy if getattr(plan, "nty", 1) > 1 else ny // 2
    cj = plan.Vx if getattr(plan, "ntx", 1) > 1 else nx // 3
    i0, j0 = max(0, min(ny - w, ci - w // 2)), max(0, min(nx - w, cj - w // 2))
    win = (i0, i0 + w + (ny % 2), j0, j0 + w + (nx % 2))        # crop parity = DEM parity
    if 2 * reach + w >= min(ny, nx):
        return {"ok": None, "note": "DEM too small for a windowed check"}
    t0 = time.time()
    a_st, s_st = orc.snr_stack_window(g._griddata, float(g._georef_info.dx), float(g._georef_info.dy), kind, scale,
                                      params, angles, win, reach, pool=pool)
    T = len(params) * len(angles)
    h, wd = win[1] - win[0], win[3] - win[2]
    sub = tuple(np.asarray(r)[win[0]:win[1], win[2]:win[3]] for r in res)
    P = orc.PARITY
    chk = orc.check_fold(sub, a_st.reshape(T, h, wd), s_st.reshape(T, h, wd), np.repeat(params, len(angles)),
                         np.tile(angles, len(params)), tie_rtol=orc.tie_window(method, kind),
                         amp_tol=(P["amp"][0], P["amp"][1] * np.max(np.abs(a_st))),
                         snr_tol=(orc.snr_tolerance(kind)[0], orc.snr_tolerance(kind)[1] * np.max(s_st)))
    out = {"ok": chk["n_bad"] == 0, "window": list(win), "templates": T, "cells": chk["n"], "bad": chk["n_bad"],
           "cells_off_the_oracle_argmax": chk["n_inexact"], "cells_below_abs_tolerance": chk["n_below_only"],
           "exact_argmax_frac": round(chk["exact_frac"], 6), "near_tie_cells": chk["n_tie"],
           "max_rel_snr_err": float("%.3g" % chk["snr_err"]), "max_rel_amp_err": float("%.3g" % chk["amp_err"]),
           "tie_rtol": orc.tie_window(method, kind), "oracle_s": round(time.time() - t0, 1)}
    if also is not None:                     # a second result on the same window and stacks (the exact mode's)
        sub2 = tuple(np.asarray(r)[win[0]:win[1], win[2]:win[3]] for r in also)
        chk2 = orc.check_fold(sub2, a_st.reshape(T, h, wd), s_st.reshape(T, h, wd), np.repeat(params, len(angles)),
                              np.tile(angles, len(params)), tie_rtol=orc.tie_window(method, kind),
                              amp_tol=(P["amp"][0], P["amp"][1] * np.max(np.abs(a_st))),
                              snr_tol=(orc.snr_tolerance(kind)[0], orc.snr_tolerance(kind)[1] * np.max(s_st)))
        out["also"] = {"bad": chk2["n_bad"], "cells_off_the_oracle_argmax": chk2["n_inexact"]}
    return out


def so_sha256():
    from scarplet_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def measured_traffic(default_workload):
    """PMC bytes per launch of every kernel slot from profiles/traffic.json ({slot: bytes}) -
    only if that file was measured (tools/prof_run.sh) on this very library."""
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if not (default_workload and os.path.exists(tf)):
        return None
    try:
        t = json.load(open(tf))
        if t.get("so_sha256") != so_sha256():
            return None
        return t.get("bytes_per_launch", {})
    except Exception:
        return None


def other_mode_leg(step, ctx, a, units, like_main=False):
    """The same workload in the mode the headline is NOT timed in (the float32 search when the line is exact, and the other
    way round): a timed loop of its own after the headline's, reported beside it.  ``like_main``: the same steps and warm-up
    as the line itself (the small configs: milliseconds); else two steps after one untimed one (C3: seconds each).  Never
    costs the line."""
    import copy
    try:
        b = copy.copy(a)
        if not like_main:
            b.steps, b.warmup, b.warmup_seconds = min(a.steps, 2), 1 if a.steps > 2 else 0, 0.0
        dt, _ = timed_loop(step, ctx, b, None)
        return {"value": round(units / (dt / b.steps) / 1e6, 1), "unit": "Mpx·template/s",
                "ms_per_step": round(1e3 * dt / b.steps, 2 if dt / b.steps >= 0.01 else 4), "steps": b.steps,
                "roofline_frac": round(units / (dt / b.steps) * ALGO_BYTES_PER_UNIT / 1e9 / HBM_PEAK_GBS, 4)}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


# ----------------------------------------------------------------------------- the other BASELINE configs
# (config, timed steps, warm-up steps at least - and 0.5 s of them, --warmup-seconds: a loop of 25 ms after the host-side
#  legs of the headline ran at the idle clock).  Timed loops of 0.15 s and more, so that the line's gpu.clock_mhz is a
#  measurement of that loop
OTHER_CONFIGS = (("C1", 400, 5), ("C2", 8, 2), ("C5", 60, 3), ("C1F", 6, 2))


def other_config_line(a, cfg, steps, warmup, device, pool):
    """One of BASELINE.json's other single-GPU configs (configs[0], [1], [4] as SURVEY.md 8d defines
    them), timed like the headline - W warm-up steps, K steps between device syncs, DEM resident -
    and verified against the oracle like the headline.  Part of the default line so that these
    numbers are driver-run, not builder-run."""
    import copy
    import scarplet_amd as sl
    b = copy.copy(a)
    b.config, b.n, b.ages, b.angles, b.steps, b.warmup, b.group = cfg, 0, 0, 0, steps, warmup, 0
    g, Template, scales, params, angles, label, kind = workload(b)
    ny, nx = g._griddata.shape
    units = float(ny) * nx * len(params) * len(angles) * len(scales)
    m = sl.Matcher(g, device=device)
    descs = []
    for sc in scales:
        arr, bbox, area = m.describe(Template, sc, params, angles)
        plan, sp = m.plan_for(bbox, area, b.method, None, n_params=len(params))
        descs.append((arr, sp, plan, m.exact_window_for(arr, sp), m.end_twins(arr, len(params), angles)))
    settle_stats = {}

    def make_step(exact):
        def step():
            for (arr_, sp_, _, win_, twin_) in descs:      # one result set per scale (C5)
                st_ = m.run_described(arr_, sp_, win_ if exact else 0.0, twin_)
                if st_:
                    settle_stats.update(st_)
        return step
    dt, prof = timed_loop(make_step(a.mode == "exact"), m.ctx, b, None)
    ms = 1e3 * dt / steps
    value = units / (dt / steps) / 1e6
    plan = descs[-1][2]
    line = {"workload": label, "mode": a.mode, "value": round(value, 1), "unit": "Mpx·template/s", "ms_per_step": round(ms, 3),
            "steps": steps, "warmup": WARMUP_RUN[0], "warmup_seconds": b.warmup_seconds,
            "tiles": "%dx%d of %dx%d" % (plan.nty, plan.ntx, plan.Ty, plan.Tx),
            "roofline_frac": round(value * 1e6 * ALGO_BYTES_PER_UNIT / 1e9 / HBM_PEAK_GBS, 4),
            "kernels_ms_per_step": {k: round(v[1] / steps, 3) for k, v in prof.items() if v[0]}}
    if a.mode == "exact":
        line["settle"] = dict(settle_stats)            # (the last scale's counters)
        line["float32_mode"] = other_mode_leg(make_step(False), m.ctx, b, units, like_main=True)
    if not a.no_e2e and len(scales) == 1:
        # the call a user makes (C1F: the reference's flagship example, sl.match(load_carrizo(), Scarp, scale=100.)): upload,
        # curvature planes, descriptors, search, float64 result planes, D2H - the first call and the same call again
        def call():
            ex = a.mode == "exact"
            if len(params) == 1:
                return sl.match(g, Template, scale=scales[0], age=float(params[0]), ang_min=float(angles[0]),
                                ang_max=float(angles[-1]), device=device, method=b.method, exact=ex)
            if len(params) == 35 and len(angles) == 181:
                return sl.match(g, Template, scale=scales[0], device=device, method=b.method, exact=ex)
            return sl.Matcher(g, device=device).search(Template, scales[0], params, angles, method=b.method, exact=ex).result()
        secs = []
        for _ in range(2):
            m.ctx.sync()
            t1 = time.perf_counter()
            r_ = call()
            secs.append(time.perf_counter() - t1)
            del r_
        line["end_to_end_ms"] = {"first_call": round(1e3 * secs[0], 2), "repeat_call": round(1e3 * secs[1], 2),
                                 "call": "sl.match(data, Template, scale=...)" if (len(params) == 1 or len(params) == 35)
                                         else "Matcher(data).search(...).result()"}
    if pool is not None and not a.no_verify:
        # the record the timed loop's mode leaves behind (the last scale's, for C5)
        make_step(a.mode == "exact")()
        res = m.ctx.get_result(np.repeat(params, len(angles)), np.tile(angles, len(params)))
        ver = verify_window(pool, res, g, kind, scales[-1], params, angles, plan, b.method)
        line["verified"] = ver["ok"]
        line["verification"] = {k: ver[k] for k in ("window", "templates", "cells", "bad", "cells_off_the_oracle_argmax",
                                                    "max_rel_snr_err", "note") if k in ver}
    return line


# ----------------------------------------------------------------------------- clocks and power of the timed loop
class GpuTelemetry(object):
    """Shader clock and board power of one GPU while the timed loop runs, read from sysfs by a
    thread of this process (no child process, no rocm-smi: a process that holds the GPU starts
    nothing): <pci device>/hwmon/hwmon*/freq1_input (Hz) and power1_average | power1_input (uW),
    pp_dpm_sclk's starred level where there is no hwmon clock.  Box-to-box spread of the headline is
    5 %; with these two numbers in the line a slow box can be told from a regression.  Every
    field is None where the files are not there or not readable (the line is printed anyway)."""

    def __init__(self, bus_id=None, root="/sys/bus/pci/devices", period=0.05):
        import glob
        self.period, self.clk, self.pw, self._run, self._thr = period, [], [], False, None
        self.dev = None
        cands = []
        if bus_id:
            cands = [os.path.join(root, bus_id.lower()), os.path.join(root, bus_id.upper()), os.path.join(root, bus_id)]
        self.dev = next((c for c in cands if os.path.isdir(c)), None)
        hw = sorted(glob.glob(os.path.join(self.dev, "hwmon", "hwmon*"))) if self.dev else []
        self.f_clk = next((os.path.join(h, "freq1_input") for h in hw if os.path.exists(os.path.join(h, "freq1_input"))), None)
        self.f_pw = next((os.path.join(h, n) for h in hw for n in ("power1_average", "power1_input")
                          if os.path.exists(os.path.join(h, n))), None)
        self.f_dpm = os.path.join(self.dev, "pp_dpm_sclk") if self.dev and os.path.exists(os.path.join(self.dev, "pp_dpm_sclk")) else None

    @staticmethod
    def _num(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def _sclk_mhz(self):
        v = self._num(self.f_clk) if self.f_clk else None
        if v is not None:
            return v / 1e6
        if self.f_dpm:
            try:
                for line in open(self.f_dpm):
                    if "*" in line:
                        return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            except (OSError, ValueError, IndexError):
                pass
        return None

    def sample(self):
        c = self._sclk_mhz()
        p = self._num(self.f_pw) if self.f_pw else None
        if c is not None:
            self.clk.append(c)
        if p is not None:
            self.pw.append(p / 1e6)

    def start(self):
        import threading
        self.clk, self.pw, self._run = [], [], True

        def loop():
            while self._run:
                self.sample()
                time.sleep(self.period)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._run = False
        if self._thr is not None:
            self._thr.join(timeout=2)
        mean = lambda v: round(float(np.mean(v)), 1) if v else None
        return {"clock_mhz": mean(self.clk), "clock_mhz_min": round(min(self.clk), 1) if self.clk else None,
                "power_w": mean(self.pw), "samples": len(self.clk) or len(self.pw),
                "source": "sysfs %s (hwmon freq1_input / power1_average, every %d ms over the timed loop)"
                          % (self.dev, int(1e3 * self.period)) if (self.clk or self.pw) else "not readable on this box"}


def device_bus_id(ctx):
    try:
        return ctx.comm_info().get("bus_id") or None
    except Exception:
        return None


# ----------------------------------------------------------------------------- rank launcher
IPC_VAR = "HSA_ENABLE_IPC_MODE_LEGACY"


def rank_environments(n, port, base=None, ipc_legacy="0"):
    """The environment of each of the n ranks started by `--gpus n` (what torch.distributed.run
    would set): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, the loopback
    address because the container's host name may not resolve.  ipc_legacy "0": the ranks get
    HSA_ENABLE_IPC_MODE_LEGACY=0 unless the caller's environment already says something else (the
    host driver of these boxes only supports dmabuf IPC; without it RCCL across processes fails with
    hipIpcGetMemHandle: invalid argument); "env": the caller's environment untouched."""
    envs = []
    for r in range(n):
        e = dict(os.environ if base is None else base)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        if ipc_legacy == "0":
            e.setdefault(IPC_VAR, "0")
        e["SCARPLET_BENCH_IPC_ATTEMPT"] = ipc_legacy       # (recorded in the line: which attempt this run was)
        envs.append(e)
    return envs


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(cmd, n, port=None, timeout=None, ipc_legacy="0"):
    """Start n rank processes of `cmd` (a list; one per GPU, like the reference's pool that starts
    and joins its own workers, core.py:180-188) and wait for them.  Called from a parent that has
    made NO HIP call and forked nothing.  Rank 0's stdout is captured (the JSON line); the other
    ranks' stdout goes to stderr.  Returns (exit code, rank 0's stdout): the code is 0 only if
    EVERY rank exited 0; as soon as one rank fails the others are terminated (they would sit in a
    collective until its timeout)."""
    import subprocess
    port = port or free_port()
    procs = []
    for r, env in enumerate(rank_environments(n, port, ipc_legacy=ipc_legacy)):
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr))
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    t0 = time.time()
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = r
        timed_out = timeout is not None and time.time() - t0 > timeout
        if failed is not None or timed_out:
            for r, p in enumerate(procs):           # exact PIDs we started, nothing by pattern
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            if timed_out and failed is None:
                failed = -1
            break
        time.sleep(0.05)
    rd.join(timeout=10)
    text = (out0[0] if out0 else b"").decode("utf-8", "replace")
    rc = 0
    if failed is not None:
        rc = codes[failed] if failed >= 0 and codes[failed] else 1
        print("bench.py: rank %s of %d failed (exit codes %r)" % (failed if failed >= 0 else "timeout", n, codes),
              file=sys.stderr)
    return (0 if rc == 0 else (rc if rc > 0 else 1)), text      # (a rank killed by a signal: 1)


def world_or_launch(a, argv=None):
    """`--gpus N` against the environment.  Under a launcher (RANK / WORLD_SIZE set, the contract's
    torch.distributed.run): WORLD_SIZE must equal --gpus, anything else exits non-zero - a line
    that says "n_gpus": 1 for --gpus 8 must never be printed.  Without one and N > 1: this process
    becomes the launcher - before any HIP call, before the oracle pool is forked - starts N ranks
    of itself, relays rank 0's JSON line and exits with their status.  Returns (rank, world,
    local) for a process that is to do the work."""
    in_env = "RANK" in os.environ or "WORLD_SIZE" in os.environ
    if in_env:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world != a.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report one as the other "
                             "(start N ranks with `python bench.py --gpus N`, or torch.distributed.run "
                             "--nproc-per-node N ... --gpus N)" % (a.gpus, world))
        return int(os.environ.get("RANK", "0")), world, int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    if a.gpus <= 1:
        return 0, 1, 0
    argv = sys.argv[1:] if argv is None else argv
    # --ipc-legacy auto: first with HSA_ENABLE_IPC_MODE_LEGACY=0 (setdefault), and if the ranks fail, once
    # more with the environment untouched - fresh processes both times (the variable is read when a process
    # initialises the GPU); the line says which attempt produced it ("ipc": {...})
    attempts = ["0", "env"] if a.ipc_legacy == "auto" else [a.ipc_legacy]
    for k, mode in enumerate(attempts):
        rc, text = launch_ranks([sys.executable, os.path.abspath(__file__)] + list(argv), a.gpus, ipc_legacy=mode)
        if rc == 0 and any(ln.strip() for ln in text.splitlines()):
            break
        if k + 1 < len(attempts):
            print("bench.py: the ranks failed with %s=%s; trying again with the environment untouched"
                  % (IPC_VAR, "0 (default)" if mode == "0" else "as found"), file=sys.stderr)
    lines = [ln for ln in text.splitlines() if ln.strip()]
    if rc == 0 and not lines:
        print("bench.py: the ranks exited 0 but rank 0 printed nothing", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    raise SystemExit(rc)


# ----------------------------------------------------------------------------- main
LAST_TELEMETRY = {}          # clocks / power of the most recent timed loop of this process (GpuTelemetry.stop())


WARMUP_RUN = [0]            # warm-up steps the last timed_loop actually ran (--warmup, --warmup-seconds)


def timed_loop(step, ctx, a, dist, after_warmup=None, max_over_ranks=None):
    """The contract's timing: W untimed steps, then EXACTLY K steps between a device sync + rank
    barrier on both sides; the MAX over ranks.  Returns (seconds, per-kernel profile of this rank)."""
    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()
    inner = step

    for kv in (getattr(a, "opt", "") or "").split(","):
        if kv:
            ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))

    def step():
        # every step pays for its own curvature spectra: what an earlier step kept (option
        # "spectra_mb": the scales of ONE multi-scale job share them) is dropped first
        ctx.forget_spectra()
        inner()
    t_w, n_w = time.perf_counter(), 0
    # (by the clock on ONE rank only: ranks that fold over a collective must run the same number of steps)
    by_clock = getattr(a, "warmup_seconds", 0.0) if (dist is None and max_over_ranks is None) else 0.0
    while n_w < a.warmup or (n_w < 100000 and time.perf_counter() - t_w < by_clock):
        step()
        n_w += 1
        if n_w >= a.warmup:
            ctx.sync()
    WARMUP_RUN[0] = n_w
    if after_warmup:
        after_warmup()
    ctx.profile(a.prof_stride)
    tel = GpuTelemetry(device_bus_id(ctx))
    barrier()
    tel.start()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    LAST_TELEMETRY.clear()
    LAST_TELEMETRY.update(tel.stop())
    if max_over_ranks is not None:
        dt = max_over_ranks(dt)
    elif dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    prof = ctx.profile_get()
    ctx.profile(0)
    return dt, prof


def all_ranks_ok(transport, ok):
    """True only if EVERY rank says ok (gathered on rank 0, the verdict broadcast back through the launcher's
    transport): whether a sharding's set-up or its first collective failed on some rank is decided together,
    so that all ranks fall back together and none is left inside a collective the others never enter."""
    flags = transport.gather(bool(ok), 0)
    verdict = b"1" if (flags is not None and all(flags)) else b"0"
    return transport.broadcast_bytes(verdict) == b"1"


def max_over(transport, v):
    """The maximum of a float over the ranks, on every rank (the contract's max-over-ranks of the timed loop)."""
    import struct
    vals = transport.gather(float(v), 0)
    m = struct.pack("d", max(vals) if vals is not None else 0.0)
    return struct.unpack("d", transport.broadcast_bytes(m))[0]


def drop_communicator(device):
    """The RCCL communicator of this process's context on `device`, if any (sc_comm_destroy)."""
    try:
        from scarplet_amd import core as _core
        _core._context(device).comm_destroy()
    except Exception:
        pass


def build_sharding(sh, backend, a, rank, world, device, transport, g, Template, scales, params, angles):
    """Everything one sharding needs on this rank with the given transport ('rccl' or 'host'):
    {step, after_warmup, plan, ctx, part, extras()}.  Raises whatever the set-up raises (communicator
    initialisation, the first halo exchange)."""
    from scarplet_amd import dist as sd
    ny, nx = g._griddata.shape
    if sh == "orientations":
        om = sd.OrientationMatcher(rank, world, g, device=device, backend=backend, transport=transport)
        mine, sp = om.describe(Template, scales[0], params, angles, a.method, a.group or None)
        # exact mode: every rank searches with the near-tie window on, keeps its own record beside the folded one, lists what
        # it knows to lie within the window of the FOLDED record; the lists (a few megabytes) travel through the launcher's
        # transport and every rank settles the union in float64 (sc_rank_candidates / sc_settle_pairs, dist.py)
        win = om.m.exact_window_for(om._keep, sp) if a.mode == "exact" else 0.0
        twin = om.m.end_twins(om._keep, len(params), angles)

        def step():
            om.run(mine, sp, win, twin)        # reset, this rank's orientations, fold over RCCL (+ the candidates' settle)

        def after_warmup():
            om.fold_seconds = 0.0
        return {"step": step, "after_warmup": after_warmup, "plan": om.m.plan, "ctx": om.m.ctx, "mode": a.mode,
                "work": lambda: (int(om.m.plan.nty * om.m.plan.ntx), 0 if mine is None else len(mine)),
                "part": "orientation grid in %d chunks, whole DEM on every rank, records folded by two all-reduces%s" % (
                    world, "; near-tie candidates of all ranks exchanged and settled in float64 on every rank" if win else ""),
                "extra_seconds": lambda: om.fold_seconds, "gather_seconds": lambda: None,
                "settle": lambda: getattr(om, "exact_stats", None),
                "result": lambda: om.result_array()}
    dm = sd.DistMatcher(rank, world, (ny, nx), float(g._georef_info.dx), float(g._georef_info.dy),
                        device=device, backend=backend, transport=transport)
    arr, bbox, area = dm.m.describe(Template, scales[0], params, angles)
    c = dm.partition_for(bbox) if a.partition == "tiles" else dm.core()
    part = ("BASELINE config C4: %d rectangles of whole FFT tiles, halo exchange" % world) if dm.cores else \
        "BASELINE config C4: %s even grid, halo exchange" % "x".join(map(str, sd.grid_dims(world, ny, nx)))
    z_core = np.ascontiguousarray(g._griddata[c[0]:c[1], c[2]:c[3]])
    dm.load(z_core, bbox)                  # halo exchange over RCCL
    plan, sp = dm.m.plan_for(bbox, area, a.method, a.group or None, n_params=len(params))
    dm.m.params, dm.m.angles = np.asarray(params, float), np.asarray(angles, float)
    halo_s, gather_s = [0.0], [0.0]
    full_box, gather_out = [None], [None]
    # exact mode: every rank settles the near-ties of its own block in float64 (sc_settle_exact on the halo-extended block:
    # the halo covers the templates' reach) before the gather - the record that travels carries the float64 argmax
    win = dm.m.exact_window_for(arr, sp) if a.mode == "exact" else 0.0
    twin = dm.m.end_twins(arr, len(params), angles)

    def step():
        t_ = time.perf_counter()
        dm.load(z_core, bbox)              # the exchange is part of a search
        halo_s[0] += time.perf_counter() - t_
        dm.m.run_described(arr, sp, win, twin)
        # ... and so is the gather: the orientation sharding's step ends with the folded
        # record on every rank, this one's with the assembled maps on rank 0
        t_ = time.perf_counter()
        if rank == 0 and gather_out[0] is None:
            gather_out[0] = np.zeros((4, ny, nx))
        full_box[0] = dm.gather(0, out=gather_out[0])
        gather_s[0] += time.perf_counter() - t_

    def after_warmup():
        halo_s[0] = 0.0
        gather_s[0] = 0.0
    return {"step": step, "after_warmup": after_warmup, "plan": plan, "ctx": dm.m.ctx, "part": part, "mode": a.mode,
            "work": lambda: (int(plan.nty * plan.ntx), len(arr)),
            "extra_seconds": lambda: halo_s[0], "gather_seconds": lambda: gather_s[0],
            "result": lambda: np.stack(full_box[0]) if full_box[0] is not None else None}


# One GPU, BASELINE config C3 (36 tiles of 2048 x 2048, 6335 templates), round 6: 3.29 s per exact step, 3.165 s float32 -
# the per-(tile, template) cost the first multi-GPU run is read against
MS_PER_TILE_TEMPLATE = {"exact": 3290.0 / (36 * 6335), "float32": 3165.0 / (36 * 6335)}


def predicted_step(works, mode, measured_ms):
    """What every rank of a sharding has to do - (FFT tiles, templates) - and what that costs at the single-GPU rate of
    the same kernels: the slowest rank bounds the step; the exchange (halo / fold / gather) comes on top.  Printed beside
    the measurement so that the first real N-GPU run can be read against it."""
    per = MS_PER_TILE_TEMPLATE.get(mode, MS_PER_TILE_TEMPLATE["float32"])
    units = [int(t_) * int(n_) for (t_, n_) in works]
    pred = max(units) * per if units else 0.0
    return {"tiles_x_templates_per_rank": units, "tiles_per_rank": [int(t_) for (t_, _) in works],
            "templates_per_rank": [int(n_) for (_, n_) in works],
            "ms_per_tile_template_1gpu": round(per, 5), "search_ms_per_step": round(pred, 1),
            "measured_over_predicted": round(measured_ms / pred, 3) if pred > 0 else None,
            "note": "tiles x templates of the slowest rank x the one-GPU cost of a (2048 x 2048 tile, template) on the C3 search "
                    "(round 6); smaller tiles cost more per cell, and the halo exchange / fold / gather are not in it"}


def run_shardings(a, rank, world, device, dist, transport, pool, g, Template, scales, params, angles, kind, units,
                  base_line):
    """Both shardings of the N-rank search (or the one --shard names), each with its own warm-up and
    barrier-bracketed K steps; returns rank 0's JSON object (None on the other ranks).

    A run on N GPUs must not come back empty: where RCCL fails on ANY rank - the communicator's
    initialisation or the first collective, found by one untimed probe step - all ranks agree on it
    (all_ranks_ok), drop their communicators and run that sharding over the host transport instead
    (host arrays through the launcher's process group, the bring-up path of --halo host); the line then says
    "transport": "host (RCCL failed: <message>)" and rccl.nranks 0.  Same process, nothing restarted."""
    shards = ["orientations", "tiles"] if a.shard == "auto" else [a.shard]
    first_backend = "host" if a.halo in ("host", "gloo") else "rccl"
    runs = {}
    for sh in shards:
        label, built, errors = None, None, []
        for backend in ([first_backend] + (["host"] if first_backend == "rccl" else [])):
            err = None
            try:
                built = build_sharding(sh, backend, a, rank, world, device, transport, g, Template, scales, params, angles)
                built["step"]()                        # the probe: the sharding's first collective, untimed
            except Exception as e:
                import traceback
                traceback.print_exc()
                err = "%s: %s" % (type(e).__name__, e)
            msgs = [m_ for m_ in (transport.gather(err, 0) or []) if m_]
            if all_ranks_ok(transport, err is None):
                label = backend if not errors else "host (RCCL failed: %s)" % errors[0]
                break
            if rank == 0:
                errors.append(msgs[0] if msgs else "a rank failed")
            else:
                errors.append(err or "another rank failed")
            drop_communicator(device)                  # the communicator of a half-built attempt must go
            built = None
        try:
            if built is None:
                raise RuntimeError("; ".join(errors) or "set-up failed")
            dt, prof = timed_loop(built["step"], built["ctx"], a, dist, built["after_warmup"],
                                  max_over_ranks=lambda v: max_over(transport, v))
            ctx, plan = built["ctx"], built["plan"]
            # what RCCL itself says about the communicator this sharding ran on, from every rank
            infos = transport.gather(ctx.comm_info(), 0)
            extra = transport.gather(built["extra_seconds"]() / a.steps, 0)
            gs = built["gather_seconds"]()
            gath = transport.gather(gs / a.steps, 0) if gs is not None else None
            tels = transport.gather(dict(LAST_TELEMETRY), 0)
            works = transport.gather(built["work"]() if "work" in built else None, 0)
            if rank == 0:
                ms = 1e3 * dt / a.steps
                line = base_line(units / (dt / a.steps) / 1e6, ms, plan, "%d (%s)" % (world, built["part"]), prof, world)
                line["transport"] = label
                line.setdefault("config", {})["mode"] = built.get("mode", getattr(a, "mode", "float32"))
                line["rccl"] = {"nranks": int(infos[0]["nranks"]),
                                "devices": [{"rank": i_["rank"], "device": i_["device"], "bus_id": i_["bus_id"]} for i_ in infos],
                                "note": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice of every rank's communicator "
                                        "(sc_comm_info); nranks 0 = no RCCL communicator (host transport)"}
                line["gpu_per_rank"] = tels
                if built.get("settle") and built["settle"]():
                    line["settle"] = built["settle"]()
                if works and all(w_ is not None for w_ in works):
                    line["predicted"] = predicted_step(works, built.get("mode", "float32"), ms)
                line["ipc"] = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get(IPC_VAR),
                               "attempt": os.environ.get("SCARPLET_BENCH_IPC_ATTEMPT", "launcher's environment")}
                key = "fold_ms" if sh == "orientations" else "halo_exchange_ms"
                line[key] = {"min_over_ranks": round(1e3 * min(extra), 3), "max_over_ranks": round(1e3 * max(extra), 3),
                             "note": "wall time per step inside the collective on a rank (exact mode: plus the candidates' "
                                     "exchange and the float64 settle of the union); the minimum is the rank that "
                                     "arrived last, i.e. the collective itself" if sh == "orientations" else
                                     "upload of the rank's core, pack, grouped ncclSend/ncclRecv, unpack, per step"}
                if gath is not None:
                    line["gather_ms"] = {"min_over_ranks": round(1e3 * min(gath), 3), "max_over_ranks": round(1e3 * max(gath), 3),
                                         "note": "inside the step: every rank's float32 record (12 B per cell) to rank 0, "
                                                 "converted there to the four float64 planes (sc_gather_result: grouped "
                                                 "ncclSend/ncclRecv, one conversion kernel per rank, D2H)"}
                line["distinct_devices"] = len({(i_["device"], i_["bus_id"]) for i_ in infos})
                if pool is not None:
                    ver = verify_window(pool, built["result"](), g, kind, scales[-1], params, angles, plan, a.method)
                    line["verified"], line["verification"] = ver["ok"], ver
                runs[sh] = line
            dist.barrier()
        except Exception as e:                         # a sharding that fails on this node must not cost the other its line
            import traceback
            traceback.print_exc()
            if rank == 0:
                runs[sh] = {"error": "%s: %s" % (type(e).__name__, e), "ms_per_step": None}
    if rank != 0:
        return None
    # the top-level line: the faster sharding among those that ran in the mode asked for (both shardings have an exact
    # mode since ABI 9)
    def rank_key(k):
        ms_ = runs[k]["ms_per_step"]
        in_mode = "error" not in runs[k] and runs[k].get("config", {}).get("mode") == getattr(a, "mode", "float32")
        return (ms_ is None, not in_mode, ms_ if ms_ is not None else float("inf"))
    first = min(runs, key=rank_key)
    out = runs[first]
    if "error" in out:
        # nothing ran: the line is still printed - value null, the errors in it - and the exit code says so
        out = {"metric": "Mpixel·template/s (DEM pixels × ages × orientations / s)", "value": None,
               "unit": "Mpx·template/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
               "error": "every sharding failed on both transports", "shardings": runs}
        out["failed"] = True
        return out
    out["sharding"] = first
    for k, v in runs.items():
        if k != first:
            out["c4_tiles" if k == "tiles" else "orientations"] = v
    return out




def roofline_block(value, world, prof, units, steps, method, default_workload):
    """`frac` is the WHOLE path against the 28-B HBM roofline (value x 28 B / 8 TB/s per GPU): the
    number that means something.  The prescribed per-kernel formula (28 B x the px.templates one
    launch of the dominant kernel serves / its mean duration) is kept as formula_frac: it credits
    one kernel with the whole path's bytes and exceeds 1 once that kernel is about half of the
    step.  traffic = PMC bytes per launch of that kernel, traffic_ratio = PMC bytes of ALL
    kernels per step / (28 B x units) - both only when profiles/traffic.json was measured on
    this very library (tools/prof_run.sh)."""
    dom = max(prof, key=lambda k: prof[k][1])
    launches, total_ms = prof[dom]
    per_launch_units = units * steps / max(launches, 1) / (world if world > 1 else 1)
    avg_s = total_ms / 1e3 / max(launches, 1)
    formula = ALGO_BYTES_PER_UNIT * per_launch_units / avg_s / 1e9 if avg_s > 0 else 0.0
    achieved = value * 1e6 * ALGO_BYTES_PER_UNIT / 1e9 / world
    r = {"bound": "hbm", "kernel": KERNEL_SYMBOLS.get(dom, dom) if method == "fft" else dom,
         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4),
         "traffic": None, "launches": int(launches), "avg_launch_us": round(1e6 * avg_s, 2),
         "formula_achieved": round(formula, 1), "formula_frac": round(formula / HBM_PEAK_GBS, 4),
         "note": "achieved / frac: the whole search, value x 28 B per px.template per GPU, against 8 TB/s. "
                 "formula_*: the prescribed per-kernel form (28 B x the px.templates of one launch of the "
                 "dominant kernel / its mean duration, HIP events on the context's stream) - it credits one "
                 "kernel with the whole path's bytes. kernel_hbm_frac: that kernel's PMC-measured bytes per "
                 "launch / its duration / 8 TB/s; traffic_ratio: PMC bytes of all kernels per step / (28 B x units)"}
    t = measured_traffic(default_workload)
    if t and t.get(dom) and avg_s > 0:
        r["traffic"] = t[dom]
        r["kernel_hbm_frac"] = round(t[dom] / avg_s / (HBM_PEAK_GBS * 1e9), 4)
        per_step = sum(t.get(k, 0) * prof[k][0] / steps for k in prof if prof[k][0])
        # (k_settle: one bracket per sc_settle_exact call - a dozen small kernels, 2 % of the exact step - is not in the table)
        if all(k in t for k in prof if prof[k][0] and k != "k_settle"):
            r["traffic_ratio"] = round(per_step / (ALGO_BYTES_PER_UNIT * units), 4)
            r["traffic_bytes_per_step"] = int(per_step)
    return r


def main():
    a = parse()
    rank, world, local = world_or_launch(a)      # --gpus N > 1 without a launcher: start the N ranks, relay, exit
    # stdout carries exactly ONE line, the JSON: whatever libraries print on the way (gloo announces
    # its ranks on stdout, RCCL its version) goes to stderr - file descriptor 1 points at stderr until
    # emit() restores it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(obj, ensure_ascii=False), flush=True)
        os.dup2(2, 1)
    dist = None
    transport = None

    import scarplet_amd as sl
    from scarplet_amd import _lib
    from scarplet_amd import dist as sd

    for kv in (a.tile_penalty or "").split(","):
        if kv:
            from scarplet_amd import _plan as _pl
            k_, v_ = kv.split("=")
            if k_.startswith("y"):                                  # y512=1.5: the column length's weights only
                _pl.TILE_PENALTY_Y[int(k_[1:])] = float(v_)
            else:
                _pl.TILE_PENALTY[int(k_)] = _pl.TILE_PENALTY_Y[int(k_)] = float(v_)
    g, Template, scales, params, angles, label, kind = workload(a)     # same seed on every rank
    ny, nx = g._griddata.shape
    if a.emulate_ranks:
        a.mode = "float32"                     # (the one-GPU emulations time the float32 search of every chunk / block)
    pool = None
    if rank == 0 and not a.emulate_ranks and not (a.no_verify and (a.no_cpu_baseline or world > 1)):
        pool = make_pool(g, kind, scales[0])   # forked before the first HIP call below (and before
                                               # the process group's threads exist)
    if world > 1:
        import torch.distributed as dist
        import datetime
        # (a rank that dies must not leave the others in a collective for gloo's default half hour)
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=900))
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from torch_transport import TorchTransport
        transport = TorchTransport()
    ndev = max(1, _lib.load().sc_device_count())
    if world > ndev and a.halo == "rccl":
        # two ranks on one device cannot form an RCCL communicator, and a line that calls them N GPUs
        # would be a lie: only the bring-up transport (--halo host: host arrays over gloo) may wrap
        raise SystemExit("bench.py: --gpus %d but %d device(s) visible; RCCL needs one GPU per rank "
                         "(bring-up on fewer devices: --halo host)" % (world, ndev))
    device = local % ndev                      # one rank per GPU; wraps only in bring-up runs (--halo host)
    n_templates = len(params) * len(angles) * len(scales)
    units = float(ny) * nx * n_templates               # px.template per step
    default_workload = (world == 1 and not a.emulate_ranks and a.config == "C3" and label.startswith("C3:")
                        and a.method == "fft")

    def base_line(value, ms, plan, ranks_label, prof, n_gpus):
        return {
            "metric": "Mpixel·template/s (DEM pixels × ages × orientations / s)",
            "value": round(value, 1), "unit": "Mpx·template/s", "n_gpus": n_gpus,
            "steps": a.steps, "warmup": a.warmup, "warmup_steps_run": WARMUP_RUN[0],
            "warmup_seconds": (a.warmup_seconds if world == 1 else 0.0), "ms_per_step": round(ms, 2 if ms >= 10 else 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if a.config in ("C2", "C3") else "reference sample DEM (tests/golden)",
            "config": {"workload": label, "method": a.method, "mode": a.mode,
                       "tiles": ("%dx%d of %dx%d" % (plan.nty, plan.ntx, plan.Ty, plan.Tx)) if a.method == "fft" else "-",
                       "group": int(getattr(plan, "group", 0)), "ranks": ranks_label},
            "roofline": roofline_block(value, n_gpus, prof, units, a.steps, a.method, default_workload),
            "kernels_ms_per_step": {k: round(v[1] / a.steps, 2) for k, v in prof.items() if v[0]},
            # shader clock and board power of this rank's GPU over the timed loop (sysfs): box spread vs regression
            "gpu": dict(LAST_TELEMETRY),
            # which sources the library that ran was compiled from (sc_build_id) and the file's own hash
            "library": {"build_id": _lib.load().sc_build_id().decode(), "so_sha256": so_sha256()[:16],
                        "abi": int(_lib.load().sc_abi_version())},
        }

    # ------------------------------------------------------------------ N > 1: both shardings
    if world > 1:
        out = run_shardings(a, rank, world, device, dist, transport, pool, g, Template, scales, params, angles,
                            kind, units, base_line)
        if rank == 0:
            if pool is not None:
                pool.terminate()
                pool.join()
            emit(out)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0 and out.get("failed"):
            raise SystemExit(1)
        return

    # ------------------------------------------------------------------ N = 1
    emu = None
    shard = "orientations" if a.shard == "auto" else a.shard
    if a.emulate_ranks > 1 and shard == "orientations":
        # the R orientation chunks of the R-rank search, one after the other on this GPU
        R = a.emulate_ranks
        om = sd.OrientationMatcher(0, 1, g, device=device)
        m = om.m
        n_par = len(params)
        arr, bbox, area = m.describe(Template, scales[0], params, angles, id_of=lambda ia, ib: ib * n_par + ia)
        plan, sp = m.plan_for(bbox, area, a.method, a.group or None, n_params=n_par)
        chunks = sd.orientation_chunks(len(angles), R)
        subs = [(_lib.sc_template * ((b1 - b0) * n_par)).from_buffer(arr, b0 * n_par * _lib.C.sizeof(_lib.sc_template))
                if b1 > b0 else None for (b0, b1) in chunks]
        per_block = np.zeros(R)
        part_label = "%d chunks of the orientation grid, whole DEM each" % R

        def step():
            for r in range(R):
                t0 = time.perf_counter()
                m.ctx.reset_best()
                if subs[r] is not None:
                    m.ctx.match(subs[r], sp, sync=True)
                per_block[r] += time.perf_counter() - t0
        ctx = m.ctx
        emu = (R, part_label, chunks, per_block, [12 * ny * nx] * R, [plan] * R)
    elif a.emulate_ranks > 1:
        # the R blocks of the R-rank search, one after the other on this GPU
        R = a.emulate_ranks
        m = sl.Matcher(device=device)
        m.ny, m.nx, m.de = ny, nx, float(g._georef_info.dx)
        arr, bbox, area = m.describe(Template, scales[0], params, angles)
        halo = sd.halo_for_search(bbox, ny, nx)
        py, px = sd.grid_dims(R, ny, nx)
        cores = sd.tile_cores(R, ny, nx, bbox) if a.partition == "tiles" else None
        lay = sd.Layout(ny, nx, py, px, halo, cores=cores)
        part_label = ("%d rectangles of whole FFT tiles" % R) if cores else "%dx%d even grid" % (py, px)
        blocks = [np.ascontiguousarray(sd.assemble_block_reference(g._griddata, lay, r)) for r in range(R)]
        halo_bytes = [8 * (blocks[r].size - (lay.core(r)[1] - lay.core(r)[0]) * (lay.core(r)[3] - lay.core(r)[2]))
                      for r in range(R)]
        per_block = np.zeros(R)
        plans = []

        def step():
            for r in range(R):
                t0 = time.perf_counter()
                m.set_block(blocks[r], lay.block_origin(r), (ny, nx), lay.core(r),
                            float(g._georef_info.dx), float(g._georef_info.dy))
                plan, sp = m.plan_for(bbox, area, a.method, a.group or None, n_params=len(params))
                m.ctx.reset_best()
                m.ctx.match(arr, sp, sync=True)
                per_block[r] += time.perf_counter() - t0
                if len(plans) < R:
                    plans.append(plan)
        ctx = m.ctx
        emu = (R, part_label, [lay.core(r) for r in range(R)], per_block, halo_bytes, plans)
    else:
        m = sl.Matcher(g, device=device)
        descs = []
        for sc in scales:
            arr, bbox, area = m.describe(Template, sc, params, angles)
            plan, sp = m.plan_for(bbox, area, a.method, a.group or None, n_params=len(params))
            descs.append((arr, sp, plan, m.exact_window_for(arr, sp), m.end_twins(arr, len(params), angles)))
        settle_stats = {}

        def make_step(exact):
            def step():
                for (arr_, sp_, _, win_, twin_) in descs:      # one result set per scale (C5)
                    st_ = m.run_described(arr_, sp_, win_ if exact else 0.0, twin_)
                    if st_:
                        settle_stats.update(st_)
            return step
        step = make_step(a.mode == "exact")
        ctx = m.ctx
        plan = descs[-1][2]

    def after_warmup():
        if emu:
            emu[3][:] = 0.0
    dt, prof = timed_loop(step, ctx, a, None, after_warmup)

    ms = 1e3 * dt / a.steps
    value = units / (dt / a.steps) / 1e6
    if emu:
        plan = emu[5][0]
    out = base_line(value, ms, plan, "1 (whole DEM)", prof, 1)
    if emu:
        R, part_label, cores_, per_block, halo_bytes, plans = emu
        pb = per_block / a.steps
        out["emulated_ranks"] = {
            "ranks": R, "partition": part_label, "cores": [list(map(int, c)) for c in cores_],
            "block_ms": [round(1e3 * v, 1) for v in pb],
            "max_block_ms": round(1e3 * float(pb.max()), 1), "sum_block_ms": round(1e3 * float(pb.sum()), 1),
            "exchanged_bytes_per_rank": int(max(halo_bytes)),
            "tiles_per_block": ["%dx%d of %dx%d" % (p_.nty, p_.ntx, p_.Ty, p_.Tx) for p_ in plans],
            "predicted_value_at_%d_gpus" % R: round(units / float(pb.max()) / 1e6, 1),
            "note": ("PREDICTED, not measured: every rank's chunk of the orientation grid searched alone on "
                     "one GPU (whole DEM); the slowest chunk bounds the %d-GPU step, the fold of the records "
                     "(exchanged_bytes_per_rank all-reduced over xGMI: a 64-bit key and a float per cell) "
                     "comes on top" % R) if shard == "orientations" else
                    ("PREDICTED, not measured: every block (core + torus halo, upload and curvature planes "
                     "included) searched alone on one GPU; the slowest block bounds the %d-GPU step, the halo "
                     "exchange (exchanged_bytes_per_rank over xGMI) comes on top" % R)}
    else:
        res = None
        if not a.no_e2e and len(scales) == 1:
            # the whole call a user makes: upload of z, curvature planes, descriptors, search,
            # float64 result planes, D2H
            def call():
                ex = a.mode == "exact"
                if default_workload:
                    return sl.match(g, Template, scale=scales[0], device=device, method=a.method, exact=ex)
                if len(params) == 1:
                    return sl.match(g, Template, scale=scales[0], age=float(params[0]), ang_min=float(angles[0]),
                                    ang_max=float(angles[-1]), device=device, method=a.method, exact=ex)
                return sl.Matcher(g, device=device).search(Template, scales[0], params, angles, method=a.method,
                                                           group=a.group or None, exact=ex).result()
            # Two calls right after the timed loop (the GPU still at its working clocks), BOTH reported as peers:
            # `value` / `seconds` is the FIRST - what a single sl.match of a fresh process pays, the 32 bytes per
            # cell of its result faulted in as fresh host pages under the copy; `repeat_call_seconds` is the same
            # call again after the first result was dropped - as in a multi-scale job - whose result block comes
            # back recycled (scarplet_amd/_hostpool.py).  The second call's result is what gets verified below:
            # the array the public API returned, not a copy fetched behind its back.
            secs = []
            for k_ in range(2):
                ctx.sync()
                t1 = time.perf_counter()
                r_ = call()
                secs.append(time.perf_counter() - t1)
                if k_ == 0:
                    del r_
            res = r_
            # where the time outside the search goes, stage by stage (by hand)
            st = {}
            t1 = time.perf_counter(); m2 = sl.Matcher(g, device=device); st["upload_and_digest"] = time.perf_counter() - t1
            t1 = time.perf_counter(); d_ = m2.describe(Template, scales[0], params, angles); st["describe"] = time.perf_counter() - t1
            del d_
            e2e = secs[0]
            out["end_to_end"] = {"value": round(units / e2e / 1e6, 1), "unit": "Mpx·template/s",
                                 "seconds": round(e2e, 3), "repeat_call_seconds": round(secs[1], 3),
                                 "overhead_ms": round(1e3 * e2e - ms, 1),
                                 "repeat_overhead_ms": round(1e3 * secs[1] - ms, 1),
                                 "stages_ms": {k: round(1e3 * v, 1) for k, v in st.items()},
                                 "call": "sl.match(data, Template, scale=...)" if (default_workload or len(params) == 1)
                                         else "Matcher(data).search(...).result()",
                                 "includes": "H2D of the float64 DEM and its digest on the device, curvature planes, "
                                             "template descriptors, search, float64 (4,ny,nx) result conversion and D2H. "
                                             "seconds: the first call (fresh result pages); repeat_call_seconds: the "
                                             "same call again, result block recycled"}
        rx = None
        if a.mode == "exact":
            out["settle"] = dict(settle_stats)
        if default_workload:
            other = "float32" if a.mode == "exact" else "exact"
            out[other + "_mode"] = other_mode_leg(make_step(a.mode != "exact"), ctx, a, units)
            if not a.no_e2e and not a.no_verify:
                # the public call in the other mode, verified on the same window and stacks as the default call's result
                rx = sl.match(g, Template, scale=scales[0], device=device, method=a.method, exact=(a.mode != "exact"))
        if not a.no_verify:
            # the result of the public call above; without it, the record the timed loop left behind
            # (the last scale's, for C5)
            if res is None:
                step()                                  # (the timed mode's record, whatever ran in between)
                res = ctx.get_result(np.repeat(params, len(angles)), np.tile(angles, len(params)))
            ver = verify_window(pool, res, g, kind, scales[-1], params, angles, plan, a.method, also=rx)
            if "also" in ver:
                out[("float32" if a.mode == "exact" else "exact") + "_mode"]["verification"] = ver.pop("also")
            out["verified"] = ver["ok"]
            out["verification"] = ver
            out["verification"]["of"] = "the arrays sl.match returned" if "end_to_end" in out else "the timed loop's record"
        del res, rx
        if default_workload and not a.no_other_configs:
            # (after the end-to-end call: these load their own DEMs into the same context)
            oc = {}
            t_oc = time.time()
            for cfg, k_, w_ in OTHER_CONFIGS:
                if time.time() - t_oc > a.other_configs_budget:   # bounded: the default run must finish within minutes
                    oc[cfg] = {"skipped": "other_configs time budget of %.0f s spent" % a.other_configs_budget}
                    continue
                try:
                    oc[cfg] = other_config_line(a, cfg, k_, w_, device, pool)
                except Exception as e:                 # must not cost the headline its line
                    import traceback
                    traceback.print_exc()
                    oc[cfg] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["other_configs"] = oc
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pool, g, params, angles)
    if pool is not None:
        pool.terminate()
        pool.join()
    emit(out)


if __name__ == "__main__":
    main()
