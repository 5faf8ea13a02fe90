#!/usr/bin/env python3
"""Soak: many scans through the Python classes (a fresh Grid per scan, pipelined feed) - the step time must not
drift and neither host nor device memory may grow once the pools are warm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import psutil
import octreelib_amd as oa
from octreelib_amd import MaxPoints, synthetic, _native as nat
from octreelib_amd.grid import Grid, GridConfig

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
scans = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
clouds = [synthetic.planar_cloud(n - 1000 * j, (20, 20, 20), seed=1, stream=j) for j in range(4)]   # sizes differ
stage = [oa.pinned_empty((n, 3)), oa.pinned_empty((n, 3))]
ctx = nat.get_context()
proc = psutil.Process()
hip = C.CDLL("libamdhip64.so")


def dev_free():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value


def put(i):
    c = clouds[i % 4]
    stage[i & 1][: len(c)] = c
    return oa.upload_async(stage[i & 1][: len(c)])


nxt = put(0)
marks = []
t0 = time.perf_counter()
for i in range(scans):
    cur = nxt
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, cur)
    nxt = put(i + 1)
    grid.subdivide([MaxPoints(64)])
    np.random.seed(0)
    grid.map_leaf_points_cuda_ransac()
    kept = grid.n_points(0)
    if i % 97 == 0:
        grid.n_leaves(0)          # (now and then a query that resolves the lazy bookkeeping)
    grid._forest.close()
    cur.release()
    if (i + 1) % (scans // 6) == 0:
        t1 = time.perf_counter()
        marks.append(((t1 - t0) * 1e3 / (scans // 6), proc.memory_info().rss / 2**20, dev_free() / 2**20, kept))
        t0 = t1
nxt.wait(); nxt.release()
for m in marks:
    print("ms/scan %.3f  host RSS %.0f MiB  device free %.0f MiB  kept %d" % m)
rss = [m[1] for m in marks]; free = [m[2] for m in marks]
assert rss[-1] - rss[1] < 64, "host memory grows"
assert free[1] - free[-1] < 64, "device memory grows"
print("soak ok")
