#!/usr/bin/env python3
"""SLAM-like use of the drop-in API: poses arrive one at a time; the first subdivide fixes the scheme,
later poses inherit it (octree_manager.py:161-171), a final subdivide + RANSAC over all poses."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd import synthetic
from octreelib_amd.grid import Grid, GridConfig

P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
clouds = [synthetic.planar_cloud(n, (16, 16, 16), seed=1, stream=p) for p in range(P)]
g0 = Grid(GridConfig(voxel_edge_length=1)); g0.insert_points(0, clouds[0][:1000]); g0.n_points(0)  # warm-up
grid = Grid(GridConfig(voxel_edge_length=1))
t0 = time.perf_counter(); grid.insert_points(0, clouds[0]); grid.subdivide([lambda p: len(p) > 64]); c = grid.n_leaves(0)
print("pose 0 insert + subdivide: %.1f ms (%d leaves)" % ((time.perf_counter() - t0) * 1e3, c))
from octreelib_amd import _native as _nat
_ctx = _nat.get_context()
for p in range(1, P):
    _ctx.sync(); _ctx.set_profiling(True)
    t0 = time.perf_counter(); grid.insert_points(p, clouds[p]); c = grid.n_leaves(p)
    dt = (time.perf_counter() - t0) * 1e3
    _ctx.sync(); tm = _ctx.timings(); _ctx.set_profiling(False)
    dev = " ".join("%s %.3f" % (k, v[0]) for k, v in sorted(tm.items()))
    print("pose %d insert (inherits the scheme) + n_leaves: %.1f ms (%d leaves)   device ms: %s" % (p, dt, c, dev))
t0 = time.perf_counter(); grid.subdivide([lambda p: len(p) > 64]); c = grid.n_leaves(0)
print("subdivide over all poses: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
np.random.seed(0)
t0 = time.perf_counter(); grid.map_leaf_points_cuda_ransac(); c = grid.n_points(0)
print("RANSAC over all poses: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    from octreelib_amd import _native as nat
    ctx = nat.get_context()
    grid2 = Grid(GridConfig(voxel_edge_length=1))
    grid2.insert_points(0, clouds[0])
    grid2.subdivide([lambda p: len(p) > 64])
    for p in range(1, P):
        grid2.insert_points(p, clouds[p])
    grid2.n_leaves(0)
    np.random.seed(0)
    ctx.set_profiling(True)
    pr = cProfile.Profile(); pr.enable()
    grid2.map_leaf_points_cuda_ransac()
    ctx.sync()
    pr.disable()
    t = ctx.timings(); ctx.set_profiling(False)
    for k, v in sorted(t.items(), key=lambda kv: -kv[1][0]):
        print("%-20s %8.3f ms x%d" % (k, v[0], v[1]))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
