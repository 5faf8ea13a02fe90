#!/usr/bin/env python3
"""The failing case of tests/test_gpu_failures.py::test_allocation_failure_sweep[... chunked bucket], step by step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from octreelib_amd import synthetic, _native as nat
from octreelib_amd.grid import Grid, GridConfig
from oracle import octree_np as onp
from tests._util import canon_from_list
from tests.test_gpu_parity import index_map, views_table

lib, ctx = nat.load(), nat.get_context()


def arm(n):
    seen = C.c_int64(0)
    lib.octl_debug_fail_alloc(int(n), C.byref(seen))
    return seen.value


for n_pts, side in ((9000, 3), (2600, 2)):
    pts = synthetic.planar_cloud(n_pts, (side, side, side), seed=11)
    idx = index_map(pts)
    og = onp.OGrid(1)
    og.insert_points(0, pts)
    og.subdivide(24)
    want = canon_from_list(og.leaf_table(0))
    for nth, probe_between in ((1, True), (1, False), (3, False), (0, True)):
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, pts)
        f = g._forest
        arm(nth)
        try:
            g.subdivide([lambda p: len(p) > 24])
            raised = False
        except MemoryError as e:
            raised = True
        seen = arm(0)
        state = (f.epoch, f.has_scheme, f.slot_epoch)
        if raised and probe_between:
            npts = g.n_points(0)
        if raised:
            g.subdivide([lambda p: len(p) > 24])
        got = canon_from_list(views_table(g.get_leaf_points(0), idx))
        same_set = dict(got) == dict(want)
        same_order = [k for k, _ in got] == [k for k, _ in want]
        first = next((i for i, (a, b) in enumerate(zip(got, want)) if a[0] != b[0]), None)
        print(n_pts, "nth", nth, "between", probe_between, "raised", raised, "seen", seen, "state after failure", state,
              "now", (f.epoch, f.has_scheme, f.slot_epoch), "set", same_set, "order", same_order, "first diff", first,
              "uniform/fast:", f.info.n_blocks, f.info.n_internal)
        if first is not None:
            nd = f.nodes
            print("    node epochs:", np.unique(nd["epoch"], return_counts=True))
            for i in range(max(0, first - 1), min(len(got), first + 4)):
                print("    ", i, "got", np.frombuffer(got[i][0][0]), np.frombuffer(got[i][0][1]), "want",
                      np.frombuffer(want[i][0][0]), np.frombuffer(want[i][0][1]))
        f.close()
