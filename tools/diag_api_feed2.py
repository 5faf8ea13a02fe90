#!/usr/bin/env python3
"""Where does a scan wait for the upload of the next one?  Host time of every library call of one scan, with a
device synchronisation after each (so a call that makes the compute stream wait shows up at once)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat

n = 10_000_000
ctx = nat.get_context(); lib = ctx.lib
pts = bench.synthetic.planar_cloud(n, (32, 32, 32), seed=1, stream=0)
pin, dbuf = [], []
for _ in range(2):
    h, d = C.c_void_p(), C.c_void_p()
    ctx.check(lib.octl_host_alloc(ctx.handle, n * 24, C.byref(h))); C.memmove(h, nat.ptr(pts), n * 24)
    ctx.check(lib.octl_dev_alloc(ctx.handle, n * 24, C.byref(d))); pin.append(h); dbuf.append(d)
np.random.seed(0); table = np.ascontiguousarray(np.random.random((1024, 6)))
e0 = np.zeros(1, dtype=np.int32); corner = np.zeros(3)
ctx.check(lib.octl_dev_upload_async(ctx.handle, dbuf[0], pin[0], n * 24))
persistent = C.c_void_p(); ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(persistent)))
for i in range(8):
    fresh = i >= 4
    k = i & 1
    T = [time.perf_counter()]; names = []
    def mark(name):
        ctx.sync(); T.append(time.perf_counter()); names.append(name)
    if fresh:
        fh = C.c_void_p(); ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fh)))
    else:
        fh = persistent; ctx.check(lib.octl_forest_clear(fh))
    slot = C.c_int32(0); info = nat.BuildInfo(); na = C.c_int64(0)
    ctx.check(lib.octl_forest_add_pose_adopt(fh, dbuf[k], n, C.byref(slot))); mark("adopt")
    ctx.check(lib.octl_dev_upload_async(ctx.handle, dbuf[1 - k], pin[1 - k], n * 24)); T.append(time.perf_counter()); names.append("upload_async")
    ctx.check(lib.octl_forest_build(fh, 64, None, 0, 0, 0, C.byref(info))); mark("build")
    ctx.check(lib.octl_forest_ransac_all(fh, 10, nat.ptr(e0), 1, nat.ptr(table), 1024, 6, 0.01)); mark("ransac_all")
    ctx.check(lib.octl_forest_apply_mask(fh, C.byref(na))); mark("apply_mask")
    if fresh:
        lib.octl_forest_destroy(fh); T.append(time.perf_counter()); names.append("destroy")
    print("scan %d (%s forest): " % (i, "fresh" if fresh else "persistent") +
          " ".join("%s %.2f" % (nm, (b - a) * 1e3) for nm, a, b in zip(names, T[:-1], T[1:])) + " | total %.2f" % ((T[-1] - T[0]) * 1e3))
ctx.check(lib.octl_ctx_sync_uploads(ctx.handle))
