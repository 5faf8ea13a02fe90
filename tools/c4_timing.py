#!/usr/bin/env python3
"""BASELINE config 4 on one GPU: one OctreeManager cube, 64 poses x 1 M points, subdivide(len > 4096) over
the union of all poses - timed through the engine (points resident in the forest's store)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd._engine import Forest
from octreelib_amd import _native as nat

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
f = Forest(1, np.zeros(3), 1.0)
t0 = time.perf_counter()
for p in range(P):
    f.add_pose(np.random.default_rng(100 + p).random((n, 3)))
f.ctx.sync()
print("insert (host -> store): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
ctx = f.ctx
for rep in range(3):
    ctx.sync(); ctx.set_profiling(True)
    t0 = time.perf_counter(); f.subdivide(K); ctx.sync(); dt = (time.perf_counter() - t0) * 1e3
    tm = ctx.timings(); ctx.set_profiling(False)
    print("subdivide(len > %d) over %d poses x %d points: %.1f ms  (%d nodes, %d levels)  -> %.0f Mpoints/s" % (
        K, P, n, dt, f.info.n_nodes, f.info.n_levels, P * n / dt / 1e3))
    print("   " + " ".join("%s %.2f" % (k, v[0]) for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0])[:12]))
