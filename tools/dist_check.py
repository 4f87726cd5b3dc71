#!/usr/bin/env python3
"""Bring-up check of the N>1 code paths under torch.distributed, compared on rank 0 with a
single-context search: the orientation-sharded search (OrientationMatcher: must be identical
in every bit) and the tiled search (DistMatcher: identical outside the tie window).
Usage (2 ranks sharing one GPU, host-side exchange):
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
      tools/dist_check.py --halo gloo"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
import scarplet_amd as sl
from scarplet_amd import _plan, synthetic, _lib, dist as sd
sys.path.insert(0, os.path.join(ROOT, "tools"))
from torch_transport import TorchTransport

ap = argparse.ArgumentParser()
ap.add_argument("--halo", default="gloo")
ap.add_argument("--size", dest="n", type=int, default=700)
a = ap.parse_args()
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ndev = max(1, _lib.load().sc_device_count())
dev = int(os.environ.get("LOCAL_RANK", 0)) % ndev
g = synthetic.synthetic_scarp(a.n, ny=a.n - 60)
z = g._griddata
ages, angles = [3.0, 30.0, 300.0], _plan.angle_grid(-1.0, 1.0)[::10]
ref = sl.Matcher(g, device=0).search(sl.Scarp, 30, ages, angles, method="fft").result_array() if rank == 0 else None

om = sd.OrientationMatcher(rank, world, g, device=dev, backend=a.halo, transport=TorchTransport())
full = om.search(sl.Scarp, 30, ages, angles, method="fft").result_array()
if rank == 0:
    print("ranks %d, orientation chunks %s: identical to the single search in every bit: %s" % (
        world, sd.orientation_chunks(len(angles), world), bool(np.array_equal(full, ref))))
    assert np.array_equal(full, ref)
del om

dm = sd.DistMatcher(rank, world, z.shape, 1.0, 1.0, device=dev, backend=a.halo, transport=TorchTransport())
c = dm.prepare(sl.Scarp, 30, ages, angles)      # cores of whole FFT tiles where that helps, else the grid
dm.search(sl.Scarp, 30, ages, angles, np.ascontiguousarray(z[c[0]:c[1], c[2]:c[3]]), method="fft")
full = dm.gather(0)
if rank == 0:
    same = (ref[1] == full[1]) & (ref[2] == full[2])
    print("ranks %d grid %s: same params %.5f, max |d snr| where same %.3g, max rel snr diff elsewhere %.3g" % (
        world, sd.grid_dims(world, *z.shape), same.mean(), np.abs(ref[3] - full[3])[same].max(),
        (np.abs(ref[3] - full[3]) / (ref[3] + 1e-12))[~same].max() if (~same).any() else 0.0))
    assert same.mean() > 0.995
dist.barrier()
dist.destroy_process_group()
