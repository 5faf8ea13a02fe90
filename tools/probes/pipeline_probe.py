#!/usr/bin/env python3
"""Timeline of octreelib_amd.ScanPipeline(2) on 10 M-point scans: when each phase of each scan starts and ends."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import octreelib_amd as oa
from octreelib_amd import MaxPoints, synthetic

n = 10_000_000
pts = synthetic.planar_cloud(n, (32, 32, 32), seed=1)
ring = [oa.pinned_empty((n, 3)) for _ in range(7)]
for r in ring:
    r[:] = pts
np.random.seed(0)
table = np.random.random((1024, 6))
T0 = time.perf_counter()
log = []


def fit(grid, i):
    t = [time.perf_counter() - T0]
    grid.subdivide([MaxPoints(64)]); t.append(time.perf_counter() - T0)
    grid.map_leaf_points_cuda_ransac(hypotheses=table); t.append(time.perf_counter() - T0)
    k = grid.n_points(0); t.append(time.perf_counter() - T0)
    log.append((i, threading.current_thread().name, [round(x * 1e3, 2) for x in t]))
    return k


for nctx in (1, 2, 3):
    with oa.ScanPipeline(nctx) as pipe:
        list(pipe.map((ring[i % 7] for i in range(4)), fit))
        log.clear()
        T0 = time.perf_counter()
        t1 = time.perf_counter()
        kept = list(pipe.map((ring[i % 7] for i in range(12)), fit))
        ms = (time.perf_counter() - t1) * 1e3 / 12
    print("contexts", nctx, "ms per scan %.2f" % ms, "kept", set(kept))
    for row in sorted(log):
        print("   ", row)
