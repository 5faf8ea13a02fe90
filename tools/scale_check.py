#!/usr/bin/env python3
"""One step of the hot path at a large point count (default 100 M points on one GPU) + timings of
the single-cube configurations (BASELINE configs 1 and 4) through the general build path."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd import _native as nat, synthetic
from octreelib_amd._engine import Forest

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = nat.get_context()
print("== Grid, %d planar points, 64x64x80 voxels" % n, flush=True)
pts = synthetic.planar_cloud(n, (64, 64, 80), seed=1)
f = Forest(0, np.zeros(3), 1.0)
t0 = time.perf_counter(); f.add_pose(pts); ctx.sync(); print("upload %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
del pts
for rep in range(2):
    t0 = time.perf_counter(); f.subdivide(64); ctx.sync(); tb = time.perf_counter() - t0
    np.random.seed(0); table = np.random.random((1024, 6))
    t0 = time.perf_counter(); f.ransac_all(10, table, 0.01); ctx.sync(); tr = time.perf_counter() - t0
    print("rep %d: subdivide %.1f ms (%.0f Mpts/s), ransac %.1f ms, voxels %d nodes %d blocks %d levels %d" % (
        rep, tb * 1e3, n / tb / 1e6, tr * 1e3, f.info.n_voxels, f.info.n_nodes, f.info.n_blocks, f.info.n_levels), flush=True)
m = f.device_mask()
print("mask: %d of %d inliers" % (int(m.sum()), len(m)), flush=True)
f.close()

print("== C4: one cube, 64 poses x 1 M points, K = 4096 (general path)", flush=True)
f = Forest(1, np.zeros(3), 1.0)
for p in range(64):
    f.add_pose(np.random.default_rng(100 + p).random((1_000_000, 3)))
for rep in range(2):
    t0 = time.perf_counter(); f.subdivide(4096); ctx.sync(); tb = time.perf_counter() - t0
    print("rep %d: subdivide %.1f ms (%.0f Mpts/s) nodes %d blocks %d levels %d" % (rep, tb * 1e3, 64e6 / tb / 1e6, f.info.n_nodes, f.info.n_blocks, f.info.n_levels), flush=True)
f.close()

print("== C1: one cube, 100 k points, K = 32 (general path)", flush=True)
f = Forest(1, np.zeros(3), 1.0)
f.add_pose(np.random.default_rng(1234).random((100_000, 3)))
for rep in range(3):
    t0 = time.perf_counter(); f.subdivide(32); ctx.sync(); tb = time.perf_counter() - t0
    print("rep %d: subdivide %.2f ms (%.1f Mpts/s) nodes %d levels %d" % (rep, tb * 1e3, 0.1 / tb, f.info.n_nodes, f.info.n_levels), flush=True)
f.close()
