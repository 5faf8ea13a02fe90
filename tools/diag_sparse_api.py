#!/usr/bin/env python3
"""Per-kernel timings of the sparse scene, and a phase breakdown of the asynchronous Python feed."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat

ctx = nat.Context(0)
n = 10_000_000
sw = bench.Workload(ctx, ctx, 0, 1, n, (256, 256, 32), "sparse", 64, False, False)
sw.step(); ctx.sync()
ctx.set_profiling(True)
for _ in range(3):
    sw.step()
ctx.sync()
tm = ctx.timings(); ctx.set_profiling(False)
print("sparse scene, ms per step:", {k: round(v[0] / 3, 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])})
sizes = sw.leaf_sizes() if False else None
sw.insert(); sw.build()
sizes = sw.leaf_sizes()
print("blocks", len(sizes), "max", sizes.max(), ">64:", int((sizes > 64).sum()), ">255:", int((sizes > 255).sum()),
      "points in >255:", int(sizes[sizes > 255].sum()))
host = sw.host_pts
sw.close()

import octreelib_amd as oa
from octreelib_amd.grid import Grid, GridConfig
pts = bench.synthetic.planar_cloud(n, (32, 32, 32), seed=1, stream=0)
stage = [oa.pinned_empty((n, 3)), oa.pinned_empty((n, 3))]
stage[0][:] = pts; stage[1][:] = pts
def T():
    return time.perf_counter()
nxt = oa.upload_async(stage[0])
for i in range(6):
    t = [T()]
    cur = nxt
    grid = Grid(GridConfig(voxel_edge_length=1)); t.append(T())
    grid.insert_points(0, cur); t.append(T())
    nxt = oa.upload_async(stage[(i + 1) & 1]) if i < 5 else None; t.append(T())
    grid.subdivide([oa.MaxPoints(64)]); t.append(T())
    np.random.seed(0)
    grid.map_leaf_points_cuda_ransac(); t.append(T())
    kept = grid.n_points(0); t.append(T())
    grid._forest.close(); t.append(T())
    cur.release(); t.append(T())
    print("scan %d: Grid() %.2f insert %.2f upload_async %.2f subdivide %.2f ransac %.2f n_points %.2f close %.2f release %.2f | total %.2f ms" % (
        (i,) + tuple((b - a) * 1e3 for a, b in zip(t[:-1], t[1:])) + ((t[-1] - t[0]) * 1e3,)))
