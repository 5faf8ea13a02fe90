#!/usr/bin/env python3
"""One-off stress: many random blocks (sizes, noise levels, thresholds, offsets) - the operator
against the oracle, exact."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from octreelib_amd.ransac import CudaRansac
from oracle import ransac_np as rnp

bad = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 6)
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    B = 4000
    sizes = rng.integers(0, 70, B).astype(np.int32)
    sizes[rng.random(B) < 0.01] = rng.integers(256, 600)
    n = int(sizes.sum())
    scale = float(rng.choice([0.05, 0.3, 1.0, 5.0]))
    off = float(rng.choice([0.0, 17.0, 1000.0, -333.0]))
    thr = float(rng.choice([0.002, 0.01, 0.05])) * max(scale, 0.2)
    cloud = rng.random((n, 3)) * scale
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for b in range(B):
        s, e = starts[b], starts[b + 1]
        if e - s >= 3 and rng.random() < 0.8:
            a, bb = rng.uniform(-1, 1, 2)
            sig = thr * float(rng.choice([0.0, 0.2, 0.5, 1.0, 2.0]))
            cloud[s:e, 2] = a * cloud[s:e, 0] + bb * cloud[s:e, 1] + rng.normal(0, sig, e - s) if sig > 0 else a * cloud[s:e, 0] + bb * cloud[s:e, 1]
    cloud += off
    H = int(rng.choice([1024, 1024, 600, 256, 100]))
    k = 6 if seed < 6 else int(rng.choice([6, 6, 3, 4, 5, 8, 12]))
    np.random.seed(seed)
    op = CudaRansac(threshold=thr, hypotheses_number=H, initial_points_number=k)
    t0 = time.time()
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    t1 = time.time()
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, thr, details=True)
    ok = (np.array_equal(counts, o_count) and np.array_equal(index, o_index)
          and np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32)) and np.array_equal(mask, o_mask))
    print("seed", seed, "H", H, "k", k, "scale", scale, "offset", off, "thr", thr, "ok" if ok else "MISMATCH",
          "gpu %.2fs oracle %.1fs" % (t1 - t0, time.time() - t1), "full leaves %.2f" % float((o_count == sizes)[sizes >= k].mean()), flush=True)
    bad += 0 if ok else 1
print("failures:", bad)
