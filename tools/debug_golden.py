import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scarplet_oracle as orc
import scarplet_amd as sl
from scarplet_amd import _plan
z = np.load(os.path.join(ROOT, "tests/golden/ref_synthetic_dem.npy"))
gold = np.load(os.path.join(ROOT, "tests/golden/ref_synthetic_match1.npz"))["res"]
g = sl.DEMGrid.from_array(z, 1.0)
for method in ("fft",):
    res = sl.match(g, sl.Scarp, scale=100, ang_max=np.pi/2, ang_min=-np.pi/2, method=method)
    amp, age, ang, snr = res
    same = np.isclose(age, gold[1], rtol=1e-9) & (ang == gold[2])
    relsnr = np.abs(snr - gold[3]) / (gold[3] + 1e-30)
    print(method, "same param frac %.4f" % same.mean(), " max rel snr err where same: %.3g" % relsnr[same & (gold[3] > 0)].max(),
          " amp err where same %.3g" % np.abs(amp - gold[0])[same].max())
    diff = ~same
    print("  differing cells:", diff.sum(), " of which snr within 4e-3 of golden:", (diff & (np.abs(snr - gold[3]) <= 4e-3 * gold[3])).sum())
    bad = diff & ~(np.abs(snr - gold[3]) <= 4e-3 * gold[3])
    idx = np.argwhere(bad)
    print("  bad:", len(idx))
    for (i, j) in idx[:3]:
        o = orc.match_template(z, 1.0, 1.0, orc.SCARP, 100, age[i, j], ang[i, j])
        print("   (%d,%d) gpu amp %.6g age %.6g ang %.4f snr %.6g | gold amp %.6g age %.6g ang %.4f snr %.6g | oracle@gpu-choice amp %.6g snr %.6g" % (
            i, j, amp[i, j], age[i, j], ang[i, j], snr[i, j], gold[0][i, j], gold[1][i, j], gold[2][i, j], gold[3][i, j], o[0][i, j], o[3][i, j]))
        o2 = orc.match_template(z, 1.0, 1.0, orc.SCARP, 100, gold[1][i, j], gold[2][i, j])
        m = sl.Matcher(g)
        a2, s2 = m.match_template(sl.Scarp, 100, gold[1][i, j], gold[2][i, j], method=method)
        print("        at gold's choice: oracle snr %.6g  gpu snr %.6g amp %.6g" % (o2[3][i, j], s2[i, j], a2[i, j]))
