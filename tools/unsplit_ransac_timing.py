#!/usr/bin/env python3
"""RANSAC on UNSPLIT 1 m voxels (~305 points per leaf: what Grid.map_leaf_points_cuda_ransac sees when
nobody called subdivide, as in the reference's own test) - the blocks beyond 255 points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd import _native as nat, synthetic
from octreelib_amd._engine import Forest

n = 10_000_000
pts = synthetic.planar_cloud(n, (32, 32, 32), seed=1)
ctx = nat.get_context()
f = Forest(0, np.zeros(3), 1.0)
f.add_pose(pts)
f.ensure_built()
np.random.seed(0)
table = np.random.random((1024, 6))
f.ransac_all(10, table, 0.01)
ctx.sync()
ctx.set_profiling(True)
for _ in range(5):
    f.ransac_all(10, table, 0.01)
ctx.sync()
t = ctx.timings()
print({k: round(v[0] / v[1], 3) for k, v in t.items()}, "kept", int(f.device_mask().sum()))
