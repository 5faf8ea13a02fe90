#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from octreelib_amd.ransac import CudaRansac
from oracle import ransac_np as rnp
rng = np.random.default_rng(123)
for origin in (np.zeros(3), np.array([5_432_100.0, -4_321_000.0, 1_250_000.0])):
    sizes = rng.integers(1, 40, 400).astype(np.int32)
    cloud = origin + rng.random((int(sizes.sum()), 3)) * 0.5
    for thr in (0.05, 0.3):
        np.random.seed(9)
        op = CudaRansac(threshold=thr, hypotheses_number=256, initial_points_number=6)
        mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
        o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, thr, details=True)
        bad = np.nonzero((counts != o_count) | (index != o_index))[0]
        print("origin", origin[0], "thr", thr, "count mismatches", len(bad), "of", len(sizes), "plane bit mismatches",
              int((planes.view(np.uint32) != o_plane.view(np.uint32)).any(axis=1).sum()), "mask diff", int((mask != o_mask).sum()))
        for b in bad[:5]:
            print("   block", b, "n", sizes[b], "dev", counts[b], index[b], planes[b], "oracle", o_count[b], o_index[b], o_plane[b])
