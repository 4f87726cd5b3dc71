"""Synthetic scarp DEMs for benchmarks and tests.

Follows the reference's own generator (scarplet/tests/test_core.py:85-101,
``generate_synthetic_scarp``): an error-function scarp of morphologic age kt0
on a planar ramp, plus Gaussian noise; BASELINE.md section 3 fixes the
parameters used for the headline metric.
"""

import numpy as np
from scipy.special import erf

from scarplet_amd.dem import DEMGrid


def synthetic_scarp(n, seed=20260101, kt0=10.0, b=0.01, sigma=0.05,
                    theta=0.2, de=1.0, dtype=np.float32, ny=None):
    """n x n (or ny x n) grid: z = -erf(yrot / (2 sqrt(kt0))) + b*yrot + noise
    with the scarp rotated by pi/2 - theta, stored as float32 like a lidar
    GeoTIFF and promoted to float64 by DEMGrid."""
    ny = n if ny is None else ny
    x = np.linspace(-n / 2, n / 2, num=n, dtype=np.float64)
    y = np.linspace(-ny / 2, ny / 2, num=ny, dtype=np.float64)
    th = np.pi / 2 - theta
    yrot = -x[np.newaxis, :] * np.sin(th) + y[:, np.newaxis] * np.cos(th)
    z = -erf(yrot / (2 * np.sqrt(kt0))) + b * yrot
    rng = np.random.default_rng(seed)
    z += sigma * rng.standard_normal((ny, n))
    return DEMGrid.from_array(z.astype(dtype), de, de)
