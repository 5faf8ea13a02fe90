#!/usr/bin/env python3
"""Why is the asynchronous Python feed slower than the C-level pipelined loop?  Per-kernel device times of the
Python loop with and without an upload in flight."""
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

def scan(cur, nxt_src, prof):
    if prof:
        ctx.set_profiling(True)
    t0 = time.perf_counter()
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, cur)
    nxt = oa.upload_async(nxt_src) if nxt_src is not None else None
    grid.subdivide([oa.MaxPoints(64)])
    np.random.seed(0)
    grid.map_leaf_points_cuda_ransac()
    kept = grid.n_points(0)
    t1 = time.perf_counter()
    tm = ctx.timings() if prof else {}
    if prof:
        ctx.set_profiling(False)
    grid._forest.close()
    cur.release()
    return nxt, (t1 - t0) * 1e3, {k: round(v[0], 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:8]}

nxt = oa.upload_async(stage[0])
for i in range(8):
    src = stage[(i + 1) & 1] if i not in (3, 7) else None
    with_upload = src is not None
    nxt, ms, tm = scan(nxt, src, prof=i >= 2)
    if nxt is None and i < 7:
        nxt = oa.upload_async(stage[(i + 1) & 1]); nxt.wait()
    print("scan %d upload-in-flight=%s: %.2f ms  %s" % (i, with_upload, ms, tm))
# the upload alone
for rep in range(3):
    t0 = time.perf_counter(); u = oa.upload_async(stage[0]); u.wait(); t1 = time.perf_counter(); u.release()
    print("upload alone: %.2f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, n * 24 / (t1 - t0) / 1e9))
