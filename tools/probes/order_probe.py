#!/usr/bin/env python3
"""Listing order of a single-pose grid whose buckets hold more than 1024 blocks (the bucket kernel then leaves the
order to order.hip): device paths against each other and against the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from octreelib_amd import synthetic
from octreelib_amd.grid import Grid, GridConfig
from oracle import octree_np as onp
from tests._util import canon_from_list
from tests.test_gpu_parity import index_map, views_table

pts = synthetic.planar_cloud(9000, (3, 3, 3), seed=11)
idx = index_map(pts)
og = onp.OGrid(1)
og.insert_points(0, pts)
og.subdivide(24)
want = canon_from_list(og.leaf_table(0))


def run(env, prebuild=False):
    for k in ("OCTL_NO_FAST_ORDER", "OCTL_NO_BUCKET_BUILD", "OCTL_NO_BUCKET_HISTORY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    g = Grid(GridConfig(voxel_edge_length=1))
    g.insert_points(0, pts)
    if prebuild:
        g.n_points(0)      # builds the top-level voxels (K < 0): the subdivide then runs over a previous scheme
    g.subdivide([lambda p: len(p) > 24])
    got = canon_from_list(views_table(g.get_leaf_points(0), idx))
    same_set = dict(got) == dict(want)
    same_order = [k for k, _ in got] == [k for k, _ in want]
    first = next((i for i, (a, b) in enumerate(zip(got, want)) if a[0] != b[0]), None)
    f = g._forest
    print(env, "prebuild", prebuild, "leaves", len(got), len(want), "set", same_set, "order", same_order, "first diff", first,
          "blocks", f.info.n_blocks, "internal", f.info.n_internal, "levels", f.info.n_levels)
    if first is not None:
        for i in range(max(0, first - 2), min(len(got), first + 6)):
            c, e = np.frombuffer(got[i][0][0]), np.frombuffer(got[i][0][1])
            c2, e2 = np.frombuffer(want[i][0][0]), np.frombuffer(want[i][0][1])
            print("   ", i, "got", c, e, "want", c2, e2)
    g._forest.close()


run({})
run({"OCTL_NO_FAST_ORDER": "1"})
run({"OCTL_NO_BUCKET_BUILD": "1"})
run({}, True)
run({"OCTL_NO_BUCKET_HISTORY": "1"}, True)
run({"OCTL_NO_BUCKET_BUILD": "1"}, True)
