#!/usr/bin/env python3
"""One-off: RANSAC on clouds far from the origin (the screen's bound grows with |origin|) against the oracle."""
import sys
sys.path.insert(0, '.')
import numpy as np
from octreelib_amd.ransac import CudaRansac
from oracle import ransac_np as rnp

bad = 0
for off in (1e6, -3e7, 1e9, 1e12):
    rng = np.random.default_rng(int(abs(off)) % 1000)
    sizes = rng.integers(0, 80, 1500).astype(np.int32)
    sizes[::97] = 300
    n = int(sizes.sum())
    cloud = rng.random((n, 3)) * 1.0
    cloud[:, 2] = 0.3 * cloud[:, 0] - 0.2 * cloud[:, 1] + rng.normal(0, 0.006, n)
    cloud += np.array([off, -off / 3, off / 7])
    np.random.seed(1)
    op = CudaRansac(threshold=0.01, hypotheses_number=1024, initial_points_number=6)
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.01, details=True)
    ok = (np.array_equal(counts, o_count) and np.array_equal(index, o_index)
          and np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32)) and np.array_equal(mask, o_mask))
    print("offset %g: %s (kept %d of %d)" % (off, "ok" if ok else "MISMATCH", int(mask.sum()), n))
    bad += 0 if ok else 1
print("failures:", bad)
