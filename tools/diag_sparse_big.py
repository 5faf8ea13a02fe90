#!/usr/bin/env python3
"""A sparse scene in a LARGE box (2048 x 2048 x 64 voxels = 2.7e8 voxel keys, 10 M points): which build path, how fast."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat, synthetic

ctx = nat.Context(0)
n = 10_000_000
for dims, general in [(d, g) for d in ((256, 256, 32), (1024, 1024, 64), (2048, 2048, 64), (4096, 4096, 64), (8192, 8192, 64)) for g in (False, True)]:
    os.environ.pop("OCTL_NO_BUCKET_BUILD", None)
    if general:
        os.environ["OCTL_NO_BUCKET_BUILD"] = "1"
    pts = synthetic.sparse_scene(n, dims, seed=7)
    wl = bench.Workload(ctx, ctx, 0, 1, n, dims, "uniform", 64, False, False)   # (the cloud is replaced below)
    ctx.check(ctx.lib.octl_dev_upload(ctx.handle, wl.d_xyz, nat.ptr(pts), pts.nbytes))
    wl.step(); ctx.sync()
    ctx.set_profiling(True)
    for _ in range(3):
        wl.step_build_only()
    ctx.sync()
    tm = ctx.timings(); ctx.set_profiling(False)
    print(dims, "general" if general else "default", "voxels", int(wl.info.n_voxels), "leaves", int(wl.info.n_blocks), "path:", bench.build_path(set(tm)),
          "build kernels ms:", round(sum(v[0] for v in tm.values()) / 3, 3),
          {k: round(v[0] / 3, 3) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:7]})
    wl.close()
