import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic
warnings.simplefilter("ignore")
M = sl.Matcher
t_acc = {}
def wrap(name):
    f = getattr(M, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); self.ctx.sync(); t_acc[name] = t_acc.get(name, 0.0) + time.perf_counter() - t0; return r
    setattr(M, name, g)
for n in ("_score_float64", "_direct_exact", "_rescore_near_ties"):
    wrap(n)
f = np.load(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/tests/golden/dem_grandcanyon.npz")
cases = [("grandcanyon Channel 1 x 181", sl.DEMGrid.from_array(f["z"].astype(float), float(f["dx"]), float(f["dy"])), sl.Channel, 10.0, [0.1], _plan.angle_grid()),
         ("synthetic scarp 1500 x 1400, Scarp 12 x 37", synthetic.synthetic_scarp(1400, ny=1500, seed=3), sl.Scarp, 30.0, list(_plan.age_grid()[::3]), _plan.angle_grid()[::5])]
for name, g, cls, scale, params, angles in cases:
    m = sl.Matcher(g)
    m.search(cls, scale, params, angles, method="fft", exact=True).result()
    t_acc.clear()
    t0 = time.perf_counter()
    m.search(cls, scale, params, angles, method="fft", exact=True); r = m.result()
    dt = time.perf_counter() - t0
    print(name, "total %.1f ms" % (1e3 * dt), {k: round(1e3 * v, 1) for k, v in t_acc.items()}, m.exact_stats)
