#!/usr/bin/env python3
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat
import octreelib_amd as oa
from octreelib_amd.grid import Grid, GridConfig

n = 10_000_000
ctx = nat.get_context()
pts = bench.synthetic.planar_cloud(n, (32, 32, 32), seed=1, stream=0)
stage = [oa.pinned_empty((n, 3)), oa.pinned_empty((n, 3))]
stage[0][:] = pts; stage[1][:] = pts
nxt = oa.upload_async(stage[0])
for i in range(6):
    T = [time.perf_counter()]; names = []
    def mark(name, sync=True):
        if sync:
            ctx.sync()
        T.append(time.perf_counter()); names.append(name)
    cur = nxt
    grid = Grid(GridConfig(voxel_edge_length=1)); mark("Grid")
    grid.insert_points(0, cur); mark("insert")
    nxt = oa.upload_async(stage[(i + 1) & 1]); mark("upload_async", sync=False)
    f = grid._forest
    f.build(64); mark("forest.build(+membership)")
    np.random.seed(0); table = np.random.random((1024, 6))
    f.ransac_all(10, table, 0.01); mark("ransac_all")
    f.apply_device_mask(); mark("apply")
    kept = grid.n_points(0); mark("n_points")
    grid._forest.close(); mark("close")
    cur.release(); mark("release")
    print("scan %d: " % i + " ".join("%s %.2f" % (nm, (b - a) * 1e3) for nm, a, b in zip(names, T[:-1], T[1:])) + " | total %.2f" % ((T[-1] - T[0]) * 1e3))
