#!/usr/bin/env python3
"""Kernel-level timings of insert+subdivide at a large point count (default 100 M)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd import _native as nat, synthetic
from octreelib_amd._engine import Forest
import cProfile, pstats

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = nat.get_context()
pts = synthetic.planar_cloud(n, (64, 64, 80), seed=1)
f = Forest(0, np.zeros(3), 1.0)
f.add_pose(pts); ctx.sync()
del pts
f.subdivide(64); ctx.sync()
ctx.set_profiling(True)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); f.subdivide(64); ctx.sync(); tb = time.perf_counter() - t0
pr.disable()
t = ctx.timings(); ctx.set_profiling(False)
print("subdivide %.2f ms" % (tb * 1e3))
tot = 0
for k, v in sorted(t.items(), key=lambda kv: -kv[1][0]):
    print("%-20s %8.3f ms x%d" % (k, v[0], v[1])); tot += v[0]
print("sum of kernels %.2f ms" % tot)
pstats.Stats(pr).sort_stats("cumulative").print_stats(10)
