#!/usr/bin/env python3
"""Wall time of the drop-in Python API (not only the kernels) on the bench workload."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd import synthetic
from octreelib_amd.grid import Grid, GridConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pts = synthetic.planar_cloud(n, (32, 32, 32), seed=1)
t = {}
t0 = time.perf_counter(); grid = Grid(GridConfig(voxel_edge_length=1)); grid.insert_points(0, pts); t["insert_points (H2D upload)"] = time.perf_counter() - t0
t0 = time.perf_counter(); grid.subdivide([lambda p: len(p) > 64]); t["subdivide"] = time.perf_counter() - t0
t0 = time.perf_counter(); c = (grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0)); t["n_nodes/n_leaves/n_points (table fetch)"] = time.perf_counter() - t0
np.random.seed(0)
t0 = time.perf_counter(); grid.map_leaf_points_cuda_ransac(); t["map_leaf_points_cuda_ransac"] = time.perf_counter() - t0
t0 = time.perf_counter(); leaves = grid.get_leaf_points(0); t["get_leaf_points (%d leaf objects)" % len(leaves)] = time.perf_counter() - t0
t0 = time.perf_counter(); s = sum(len(v.get_points()) for v in leaves[:10000]); t["get_points of 10k leaves"] = time.perf_counter() - t0
print("counters", c, "after ransac", grid.n_points(0))
for k, v in t.items():
    print("%-48s %8.1f ms" % (k, v * 1e3))
