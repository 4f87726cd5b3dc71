#!/usr/bin/env python3
"""Would overlapping the phases of the search pay?  Two contexts on ONE GPU, each with its own stream, search
half of the C3 orientation grid each - from two host threads at once (their kernels interleave: the column
pass of one beside the row pass or the forward passes of the other) and one after the other."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scarplet_amd as sl
from scarplet_amd import _lib, _plan, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
g = synthetic.synthetic_scarp(n)
ages, angs = _plan.age_grid(), _plan.angle_grid()
halves = [angs[:90], angs[90:180]]
ms = [sl.Matcher(g, ctx=_lib.Context(0)) for _ in range(2)]
work = []
for m, a in zip(ms, halves):
    arr, bbox, area = m.describe(sl.Scarp, 100, ages, a)
    plan, sp = m.plan_for(bbox, area, "fft", None, n_params=len(ages))
    work.append((m, arr, sp))

def run(k):
    m, arr, sp = work[k]
    m.ctx.reset_best(); m.ctx.match(arr, sp, sync=True)

for k in range(2): run(k)          # warm
for rep in range(2):
    t0 = time.perf_counter(); run(0); run(1); seq = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    con = time.perf_counter() - t0
    print("180 orientations x 35 ages at %d^2: one after the other %.3f s, two streams at once %.3f s (%.1f %%)" % (n, seq, con, 100 * (con / seq - 1)), flush=True)
