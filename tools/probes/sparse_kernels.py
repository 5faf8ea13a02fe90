#!/usr/bin/env python3
"""Per-kernel timers of the sparse scene's step (secondary.sparse_scene of bench.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from octreelib_amd import _native as nat

ctx = nat.Context(0)
sw = bench.Workload(ctx, ctx, 0, 1, 10_000_000, (256, 256, 32), "sparse", 64, False, False)
for _ in range(3):
    sw.step()
ctx.sync()
ctx.set_profiling(True)
for _ in range(4):
    sw.step()
ctx.sync()
tm = ctx.timings()
ctx.set_profiling(False)
tot = 0
for k, v in sorted(tm.items(), key=lambda kv: -kv[1][0]):
    print("%-18s %8.3f ms/step x%.1f" % (k, v[0] / 4, v[1] / 4))
    tot += v[0] / 4
print("sum %.3f" % tot, "blocks", sw.info.n_blocks, "voxels", sw.info.n_voxels)
sw.close()
