#!/usr/bin/env python3
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat
import octreelib_amd as oa
from octreelib_amd.grid import Grid, GridConfig

n = 10_000_000
ctx = nat.get_context(); lib = ctx.lib
pts = bench.synthetic.planar_cloud(n, (32, 32, 32), seed=1, stream=0)
stage = [oa.pinned_empty((n, 3)), oa.pinned_empty((n, 3))]
stage[0][:] = pts; stage[1][:] = pts
np.random.seed(0); table = np.ascontiguousarray(np.random.random((1024, 6)))
e0 = np.zeros(1, dtype=np.int32)
for variant in ("python", "raw_build", "raw_build+voxels", "raw_build+slotvox", "prealloc_dev"):
    nxt = oa.upload_async(stage[0])
    pre = [oa.upload_async(stage[0]), oa.upload_async(stage[1])]
    for u in pre:
        u.wait()
    tot = []
    for i in range(5):
        t0 = time.perf_counter()
        cur = nxt
        grid = Grid(GridConfig(voxel_edge_length=1))
        grid.insert_points(0, cur)
        if variant == "prealloc_dev":
            nxt = pre[(i + 1) & 1]
            ctx.check(lib.octl_dev_upload_async(ctx.handle, nxt.ptr, nat.ptr(stage[(i + 1) & 1]), n * 24))
        else:
            nxt = oa.upload_async(stage[(i + 1) & 1])
        f = grid._forest
        if variant == "python":
            f.build(64)
        else:
            info = nat.BuildInfo()
            ctx.check(lib.octl_forest_build(f.handle, 64, None, 0, 0, 0, C.byref(info)))
            f.info = info; f.n_ord = int(info.n_points); f._dirty = False; f._invalidate(); f.epoch += 1; f.has_scheme = True
            if variant == "raw_build+voxels":
                _ = f.voxels
            if variant == "raw_build+slotvox":
                m = C.c_int64(0)
                ctx.check(lib.octl_forest_get_slot_voxels(f.handle, 0, 0, None, C.byref(m)))
        ctx.check(lib.octl_forest_ransac_all(f.handle, 10, nat.ptr(e0), 1, nat.ptr(table), 1024, 6, 0.01))
        f.apply_device_mask()
        t1 = time.perf_counter()
        grid._forest.close()
        if variant != "prealloc_dev" or i == 0:
            if cur not in pre:
                cur.release()
        tot.append((t1 - t0) * 1e3)
    ctx.check(lib.octl_ctx_sync_uploads(ctx.handle))
    print(variant, " ".join("%.2f" % t for t in tot))
