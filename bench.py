#!/usr/bin/env python3
"""Headline benchmark: Mpixel.template/s of one full template search.

    python bench.py --gpus N --steps K --warmup W

One "step" is one search of the 35-age x 181-orientation Scarp grid (scale
100) over the 10000 x 10000 synthetic DEM of BASELINE.md (config C3), DEM
already resident in HBM.  With N > 1 (launched by torch.distributed.run, one
rank per GPU) the same DEM is cut into a py x px tile grid (C4: 2 x 4), every
rank exchanges halos over RCCL and searches its tile: total work is fixed, so
scaling is "strong".  torch is used for the launcher contract only (rank
rendezvous, barrier, max-reduce of the timings); the product path is
ctypes -> libscarplet_hip.so.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_UNIT = 28.0        # bytes per px.template
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec
# profiling slot of the library -> kernel symbols as rocprofv3 lists them (Scarp / Ricker
# searches at T = 512..2048; each tile pair takes one launch of either instantiation)
KERNEL_SYMBOLS = {"k_inv_cols": "k_inv_cols_sym<T,false|true,false>", "k_inv_rows": "k_inv_rows_fast<T,false,false,false>"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", dest="n", type=int, default=10000, help="DEM size (default: C3)")
    ap.add_argument("--ages", type=int, default=35)
    ap.add_argument("--angles", type=int, default=181)
    ap.add_argument("--method", default="fft")
    ap.add_argument("--group", type=int, default=0, help="templates per inverse launch (0: auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-stride", type=int, default=16)
    ap.add_argument("--halo", default="rccl", choices=["rccl", "gloo"],
                    help="halo exchange executor for --gpus > 1 (gloo: host arrays, for bring-up)")
    return ap.parse_args()


def _cpu_one(args):
    z, age, ang = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scarplet_oracle as orc
    orc.match_template(z, 1.0, 1.0, orc.SCARP, 100, age, ang, workers=1)
    return 1


def cpu_baseline(z_full, ages, angles, budget_s=25.0):
    """The oracle (float64 FFT restatement of core.py:297-377) timed on this
    box's host cores the way the reference runs it: a process pool over
    templates (core.py:180-183), one single-threaded FFT convolution per
    process.  Bounded sample: a 2048 x 2048 crop of the DEM, one template per
    worker drawn across the (age, angle) grid."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    workers = max(1, min(cores, 64))
    c = min(2048, z_full.shape[0])
    z = np.ascontiguousarray(z_full[:c, :c], dtype=np.float64)
    rng = np.random.default_rng(0)
    picks = [(z, float(ages[rng.integers(len(ages))]), float(angles[rng.integers(len(angles))]))
             for _ in range(workers)]
    ctx = mp.get_context("fork")
    with ctx.Pool(workers) as pool:
        pool.map(_cpu_one, picks[:workers])          # warm the workers (imports, FFT plans)
        t0 = time.time()
        done = sum(pool.map(_cpu_one, picks, chunksize=1))
        dt = time.time() - t0
    return {"value": round(c * c * done / dt / 1e6, 3), "unit": "Mpx.template/s", "cores": int(workers),
            "kind": "port", "sample": "%d templates of the 35x181 grid on a %dx%d crop of the DEM, "
            "oracle/scarplet_oracle.py, process pool of %d single-threaded workers (%d cores visible), %.1f s"
            % (done, c, c, workers, cores, dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")

    import scarplet_amd as sl
    from scarplet_amd import _plan, synthetic, _lib
    from scarplet_amd import dist as sd
    ndev = max(1, _lib.load().sc_device_count())
    device = local % ndev                      # one rank per GPU; wraps only in bring-up runs

    ages = _plan.age_grid()[np.round(np.linspace(0, 34, a.ages)).astype(int)]
    angles = _plan.angle_grid()[np.round(np.linspace(0, 180, a.angles)).astype(int)]
    g = synthetic.synthetic_scarp(a.n)                 # same seed on every rank
    n_templates = len(ages) * len(angles)
    units = float(a.n) * a.n * n_templates             # px.template per step

    if world == 1:
        m = sl.Matcher(g, device=device)
        arr, bbox, area = m.describe(sl.Scarp, 100, ages, angles)
        plan, sp = m.plan_for(bbox, area, a.method, a.group or None, n_params=len(ages))

        def step():
            m.ctx.reset_best()
            m.ctx.match(arr, sp, sync=True)
        ctx = m.ctx
    else:
        dm = sd.DistMatcher(rank, world, (a.n, a.n), 1.0, 1.0, device=device, backend=a.halo)
        c = dm.core()
        z_core = np.ascontiguousarray(g._griddata[c[0]:c[1], c[2]:c[3]])
        arr, bbox, area = dm.m.describe(sl.Scarp, 100, ages, angles)
        dm.load(z_core, bbox)                          # halo exchange over RCCL
        plan, sp = dm.m.plan_for(bbox, area, a.method, a.group or None, n_params=len(ages))

        def step():
            dm.load(z_core, bbox)                      # the exchange is part of a search
            dm.m.ctx.reset_best()
            dm.m.ctx.match(arr, sp, sync=True)
        ctx = dm.m.ctx

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    ctx.profile(a.prof_stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    prof = ctx.profile_get()

    if rank == 0:
        ms = 1e3 * dt / a.steps
        value = units / (dt / a.steps) / 1e6
        # dominant kernel by total device time; its algorithmic bytes per launch
        # = 28 B x the px.templates one launch serves (DESIGN.md "Roofline")
        dom = max(prof, key=lambda k: prof[k][1])
        launches, total_ms = prof[dom]
        core = ctx.core
        core_px = (core[1] - core[0]) * (core[3] - core[2])
        per_launch_units = core_px * n_templates * a.steps / max(launches, 1)
        avg_s = total_ms / 1e3 / max(launches, 1)
        achieved = ALGO_BYTES_PER_UNIT * per_launch_units / avg_s / 1e9 if avg_s > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")
        default_workload = (world == 1 and a.n == 10000 and len(ages) == 35 and len(angles) == 181
                            and a.method == "fft")
        if os.path.exists(tf) and default_workload:      # measured on exactly this workload
            try:
                traffic = json.load(open(tf)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixel·template/s (DEM pixels × ages × orientations / s)",
            "value": round(value, 1), "unit": "Mpx·template/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d synthetic erf-scarp DEM (BASELINE.md s3), Scarp, "
                       "scale=100, %d ages x %d orientations" % (
                           "C3" if (a.n == 10000 and len(ages) == 35 and len(angles) == 181) else "reduced",
                           a.n, a.n, len(ages), len(angles)),
                       "method": a.method, "tiles": "%dx%d of %dx%d" % (plan.nty, plan.ntx, plan.Ty, plan.Tx),
                       "ranks": "%d (%s tile grid)" % (world, "x".join(map(str, sd.grid_dims(world, a.n, a.n))))},
            "roofline": {"bound": "hbm", "kernel": KERNEL_SYMBOLS.get(dom, dom) if a.method == "fft" else dom,
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "launches": int(launches),
                         "avg_launch_us": round(1e6 * avg_s, 2),
                         "pipeline_frac": round(value * 1e6 * ALGO_BYTES_PER_UNIT / (HBM_PEAK_GBS * 1e9 * world), 4)},
            "kernels_ms_per_step": {k: round(v[1] / a.steps, 2) for k, v in prof.items() if v[0]},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(g._griddata, ages, angles)
        print(json.dumps(out, ensure_ascii=False))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
