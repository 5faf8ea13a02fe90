#!/usr/bin/env python3
"""octl_forest_build of the headline scene on a FRESH forest per scan (what Grid() per scan does) against a
forest that is cleared and reused (what bench.py's headline does): wall time and the kernels' time."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import octreelib_amd as oa
from octreelib_amd import _native as nat, synthetic
from octreelib_amd._engine import Forest

n = 10_000_000
pts = synthetic.planar_cloud(n, (32, 32, 32), seed=1, stream=0)
ctx = nat.get_context()
stage = oa.pinned_empty((n, 3)); stage[:] = pts
dev = oa.upload_async(stage); dev.wait()
for mode in ("fresh", "fresh", "reused", "fresh+profiling"):
    walls = []
    f = None
    for i in range(6):
        if mode != "reused" or f is None:
            f = Forest(0, np.zeros(3), 1.0)
        else:
            ctx.check(ctx.lib.octl_forest_clear(f.handle))
        slot = C.c_int32(-1)
        ctx.check(ctx.lib.octl_forest_add_pose_adopt(f.handle, dev.ptr, n, C.byref(slot)))
        ctx.sync()
        if mode.endswith("profiling"):
            ctx.set_profiling(True)
        info = nat.BuildInfo()
        t0 = time.perf_counter()
        ctx.check(ctx.lib.octl_forest_build(f.handle, 64, None, 0, 0, 0, C.byref(info)))
        walls.append((time.perf_counter() - t0) * 1e3)
        if mode.endswith("profiling"):
            tm = ctx.timings(); ctx.set_profiling(False)
            if i == 5:
                print("   kernels:", {k: round(v[0], 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])})
        if mode != "reused":
            f.close()
    print(mode, " ".join("%.2f" % w for w in walls))

# the Python-side bookkeeping that follows a build (Forest._update_membership), piece by piece
marks = {}
for i in range(8):
    f = Forest(0, np.zeros(3), 1.0)
    f.add_pose_device(dev.ptr, n, adopt=True)
    info = nat.BuildInfo()
    t = [time.perf_counter()]
    ctx.check(ctx.lib.octl_forest_build(f.handle, 64, None, 0, 0, 0, C.byref(info))); t.append(time.perf_counter())
    f.info = info; f.n_ord = int(info.n_points); f._dirty = False; f._invalidate(); f.epoch += 1; f.has_scheme = True
    m = C.c_int64(0)
    ctx.check(ctx.lib.octl_forest_get_voxels(f.handle, 0, None, C.byref(m))); t.append(time.perf_counter())
    v = np.empty((m.value, 3), dtype=np.int64)
    ctx.check(ctx.lib.octl_forest_get_voxels(f.handle, m.value, nat.ptr(v), C.byref(m))); t.append(time.perf_counter())
    f._voxels = v
    ctx.check(ctx.lib.octl_forest_get_slot_voxels(f.handle, 0, 0, None, C.byref(m))); t.append(time.perf_counter())
    vids = np.empty(m.value, dtype=np.int32)
    ctx.check(ctx.lib.octl_forest_get_slot_voxels(f.handle, 0, m.value, nat.ptr(vids), C.byref(m))); t.append(time.perf_counter())
    f._update_membership(); t.append(time.perf_counter())
    for k, (a, b) in zip(("build", "get_voxels(count)", "get_voxels", "slot_voxels(count)", "slot_voxels(get)", "membership again (numpy + 2 calls)"), zip(t, t[1:])):
        marks.setdefault(k, []).append((b - a) * 1e3)
    f.close()
print({k: round(float(np.median(v)), 3) for k, v in marks.items()})
