#!/usr/bin/env python3
"""Differential soak of the RANSAC prescreen: the same launch with the prescreen on and off (NO_RANSAC_PRESCREEN) must
give the same count, winner index, f32 plane bits and mask for every block.  No oracle in the loop, so thousands of
blocks per second: a bound that is violated anywhere in the space below shows as a mismatch.
    python tools/prescreen_diff.py [first_seed] [last_seed]
Per seed: 6 000 blocks of 3..255 points; block geometry drawn from planes with noise / outliers, lines, point
clusters, lattices, parallel plane pairs, slivers; extents from 1e-3 to 300, offsets from 0 to 1e5, thresholds from
1e-4 to 10 of the noise scale; H in {1024, 700, 257}, k in {3, 4, 5, 6}."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from octreelib_amd import _native as nat
from octreelib_amd.ransac import CudaRansac

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 20)
# "eligible": extents, offsets and thresholds inside the prescreen's range for (nearly) every block - the bound is at
# work everywhere; default: the wide sweep, where many launches stand aside
ELIGIBLE = len(sys.argv) > 3 and sys.argv[3] == "eligible"
ctx = nat.get_context()
bad = 0
blocks = 0
t_start = time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng(10_000 + seed)
    B = 6000
    sizes = rng.integers(3, 64, B).astype(np.int32)
    big = rng.random(B) < 0.05
    sizes[big] = rng.integers(64, 256, int(big.sum()))
    n = int(sizes.sum())
    starts = np.concatenate(([0], np.cumsum(sizes)))
    if ELIGIBLE:
        scale = float(10.0 ** rng.uniform(-1.3, 1.0))
        off = float(rng.choice([0.0, 3.0, 40.0, 200.0])) * float(rng.choice([1.0, -1.0]))
        sigma = scale * float(10.0 ** rng.uniform(-2.5, -1.0))
        thr = sigma * float(10.0 ** rng.uniform(-0.3, 1.0))
    else:
        scale = float(10.0 ** rng.uniform(-3, 2.5))
        off = float(rng.choice([0.0, 3.0, 40.0, 900.0, 1.0e5])) * float(rng.choice([1.0, -1.0]))
        sigma = scale * float(10.0 ** rng.uniform(-3.5, -1.0))
        thr = sigma * float(10.0 ** rng.uniform(-0.5, 1.0))
    cloud = np.empty((n, 3))
    kind = rng.integers(0, 7, B)
    for b in range(B):
        s, e = starts[b], starts[b + 1]
        m = e - s
        base = rng.random(3) * scale * 20.0
        p = rng.random((m, 3)) * scale
        kd = kind[b]
        if kd <= 1:      # plane + noise + outliers
            a, c = rng.uniform(-1, 1, 2)
            p[:, 2] = a * p[:, 0] + c * p[:, 1] + rng.normal(0, sigma, m)
            o = rng.random(m) < rng.choice([0.0, 0.1, 0.3])
            p[o, 2] = rng.random(int(o.sum())) * scale
        elif kd == 2:    # a line
            d = rng.normal(size=3)
            p = np.outer(rng.random(m), d / np.linalg.norm(d)) * scale + rng.normal(0, sigma, (m, 3))
        elif kd == 3:    # clusters
            c = rng.random((3, 3)) * scale
            p = c[rng.integers(0, 3, m)] + rng.normal(0, sigma * 0.1, (m, 3))
        elif kd == 4:    # lattice (tied counts, distances on the threshold)
            p = rng.integers(0, 8, (m, 3)) * (scale / 8.0)
            p[:, 2] = rng.integers(0, 3, m) * thr
        elif kd == 5:    # two parallel planes
            p[:, 2] = 0.3 * p[:, 0] + (rng.random(m) < 0.5) * (2.5 * thr) + rng.normal(0, sigma * 0.3, m)
        else:            # a sliver
            p[:, 1] *= 1e-3
            p[:, 2] = 0.2 * p[:, 0] + rng.normal(0, sigma, m)
        cloud[s:e] = p + base
    cloud += off
    H = int(rng.choice([1024, 1024, 1024, 700, 257]))
    k = int(rng.choice([6, 6, 6, 3, 4, 5]))
    np.random.seed(seed)
    op = CudaRansac(threshold=thr, hypotheses_number=H, initial_points_number=k)
    res = []
    for off_switch in (0, 1):
        ctx.set_option("NO_RANSAC_PRESCREEN", off_switch)
        res.append(op.evaluate(cloud, sizes, details=True))
    ctx.set_option("NO_RANSAC_PRESCREEN", 0)
    (m0, p0, c0, i0), (m1, p1, c1, i1) = res
    ok = (np.array_equal(c0, c1) and np.array_equal(i0, i1) and np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
          and np.array_equal(m0, m1))
    blocks += B
    if not ok:
        bad += 1
        w = np.flatnonzero((c0 != c1) | (i0 != i1))
        print("seed", seed, "MISMATCH in blocks", w[:10], "sizes", sizes[w[:10]], "kinds", kind[w[:10]],
              "scale", scale, "offset", off, "thr", thr, "H", H, "k", k, flush=True)
    elif seed % 10 == 0:
        print("seed", seed, "ok  scale %.3g offset %.3g thr %.3g H %d k %d  (%d blocks, %.0f s)" %
              (scale, off, thr, H, k, blocks, time.time() - t_start), flush=True)
print("seeds", lo, "..", hi, "eligible" if ELIGIBLE else "wide", "blocks", blocks, "failures:", bad)
sys.exit(1 if bad else 0)
