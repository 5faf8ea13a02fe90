#!/usr/bin/env python3
"""Probe: aggregate throughput of two independent step sequences (two contexts = two streams, two
forests, two host threads) against one, on one GPU."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd import _native as nat, synthetic

n = 10_000_000
pts = np.ascontiguousarray(synthetic.planar_cloud(n, (32, 32, 32), seed=1))
np.random.seed(0)
table = np.ascontiguousarray(np.random.random((1024, 6)))
corner = np.zeros(3)
e0 = np.zeros(1, dtype=np.int32)

class Worker:
    def __init__(self, d_xyz=None):
        self.ctx = nat.Context(0)
        self.lib = self.ctx.lib
        if d_xyz is None:
            d_xyz = C.c_void_p()
            self.ctx.check(self.lib.octl_dev_alloc(self.ctx.handle, pts.nbytes, C.byref(d_xyz)))
            self.ctx.check(self.lib.octl_dev_upload(self.ctx.handle, d_xyz, nat.ptr(pts), pts.nbytes))
        self.d_xyz = d_xyz
        self.fh = C.c_void_p()
        self.ctx.check(self.lib.octl_forest_create(self.ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(self.fh)))
        self.info, self.slot, self.alive = nat.BuildInfo(), C.c_int32(0), C.c_int64(0)
    def step(self):
        c, l = self.ctx, self.lib
        c.check(l.octl_forest_clear(self.fh))
        c.check(l.octl_forest_add_pose_device(self.fh, self.d_xyz, n, C.byref(self.slot)))
        c.check(l.octl_forest_build(self.fh, 64, None, 0, 0, 0, C.byref(self.info)))
        c.check(l.octl_forest_ransac_all(self.fh, 10, nat.ptr(e0), 1, nat.ptr(table), 1024, 6, 0.01))
        c.check(l.octl_forest_apply_mask(self.fh, C.byref(self.alive)))
    def run(self, k):
        for _ in range(k):
            self.step()
        self.ctx.sync()

a = Worker()
b = Worker(a.d_xyz)
for w in (a, b):
    w.run(2)
K = 20
t0 = time.perf_counter(); a.run(K); t1 = time.perf_counter() - t0
print("one stream : %.2f ms per step" % (t1 / K * 1e3))
t0 = time.perf_counter()
ths = [threading.Thread(target=w.run, args=(K // 2,)) for w in (a, b)]
[t.start() for t in ths]; [t.join() for t in ths]
t2 = time.perf_counter() - t0
print("two streams: %.2f ms per step (aggregate), kept %d %d" % (t2 / K * 1e3, a.alive.value, b.alive.value))
