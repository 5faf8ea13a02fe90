#!/usr/bin/env python3
"""cProfile of the pipelined Python-API loop of bench.py (secondary.api_pipelined): where the host time of a
scan goes between the library's synchronisation points."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import octreelib_amd as oa
from octreelib_amd import MaxPoints, synthetic
from octreelib_amd.grid import Grid, GridConfig

n = 10_000_000
pts = synthetic.planar_cloud(n, (32, 32, 32), seed=1, stream=0)
stage = [oa.pinned_empty((n, 3)), oa.pinned_empty((n, 3))]
stage[0][:] = pts
stage[1][:] = pts
marks = {}


def api_loop(count, timing=False, overlap=True):
    nxt = oa.upload_async(stage[0])
    for i in range(count):
        t = [time.perf_counter()]
        cur = nxt
        grid = Grid(GridConfig(voxel_edge_length=1))
        grid.insert_points(0, cur); t.append(time.perf_counter())
        nxt = oa.upload_async(stage[(i + 1) & 1]) if i + 1 < count else None
        if nxt is not None and not overlap:
            nxt.wait()
        t.append(time.perf_counter())
        grid.subdivide([MaxPoints(64)]); t.append(time.perf_counter())
        np.random.seed(0)
        grid.map_leaf_points_cuda_ransac(); t.append(time.perf_counter())
        kept = grid.n_points(0); t.append(time.perf_counter())
        grid._forest.close()
        cur.release(); t.append(time.perf_counter())
        if timing:
            for k, (a, b) in zip(("insert", "upload_async", "subdivide", "ransac", "n_points", "close+release"), zip(t, t[1:])):
                marks.setdefault(k, []).append((b - a) * 1e3)
    return kept


api_loop(2)
t0 = time.perf_counter(); api_loop(8, True); print("ms per scan: %.2f" % ((time.perf_counter() - t0) * 1e3 / 8))
print({k: round(float(np.median(v)), 3) for k, v in marks.items()})
marks.clear()
t0 = time.perf_counter(); api_loop(8, True, overlap=False); print("upload waited for (no overlap): ms per scan: %.2f" % ((time.perf_counter() - t0) * 1e3 / 8))
print({k: round(float(np.median(v)), 3) for k, v in marks.items()})
from octreelib_amd import _native as nat
ctx = nat.get_context()
f = None
marks.clear()
nxt = oa.upload_async(stage[0])
for i in range(8):
    cur = nxt
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, cur)
    nxt = oa.upload_async(stage[(i + 1) & 1]) if i + 1 < 8 else None
    f = grid._forest
    import ctypes as C
    t = [time.perf_counter()]
    info = nat.BuildInfo()
    ctx.check(ctx.lib.octl_forest_build(f.handle, 64, None, 0, 0, 0, C.byref(info))); t.append(time.perf_counter())
    f.info = info; f.n_ord = int(info.n_points); f._dirty = False; f._invalidate(); f.epoch += 1; f.has_scheme = True
    _ = f.voxels; t.append(time.perf_counter())
    m = C.c_int64(0)
    ctx.check(ctx.lib.octl_forest_get_slot_voxels(f.handle, 0, 0, None, C.byref(m))); t.append(time.perf_counter())
    vids = np.empty(m.value, dtype=np.int32)
    ctx.check(ctx.lib.octl_forest_get_slot_voxels(f.handle, 0, m.value, nat.ptr(vids), C.byref(m))); t.append(time.perf_counter())
    f._update_membership(); t.append(time.perf_counter())
    for k, (a, b) in zip(("build", "voxels", "slot_voxels(count)", "slot_voxels(get)", "membership(rest)"), zip(t, t[1:])):
        marks.setdefault(k, []).append((b - a) * 1e3)
    ctx.sync()
    f.close(); cur.release()
print("pieces of subdivide under a concurrent upload:", {k: round(float(np.median(v)), 3) for k, v in marks.items()})
pr = cProfile.Profile(); pr.enable(); api_loop(8); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
