"""The multi-GPU code path on the one GPU a test box has: a 1-rank RCCL communicator with the
local part forced through AllGather + grouped Send/Recv (OCTL_ROUTE_SELF_SENDRECV)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_grid_single_rank_through_rccl(monkeypatch):
    from octreelib_amd import synthetic
    from octreelib_amd.distributed import ShardedGrid
    from octreelib_amd.grid import Grid, GridConfig

    monkeypatch.setenv("OCTL_ROUTE_SELF_SENDRECV", "1")
    pts = synthetic.planar_cloud(40_000, (4, 4, 4), seed=1, stream=5)
    sg = ShardedGrid(1, rank=0, n_ranks=1, comm_broadcast=lambda b: b, device=0)
    try:
        n = sg.insert_points(pts, index_base=1000)
        assert n == len(pts)
        # stable routing: received order == sent order, global indices attached
        assert np.array_equal(sg.routed_global_indices(), 1000 + np.arange(len(pts)))
        sg.subdivide(64)
        np.random.seed(0)
        table = np.random.random((1024, 6))
        sg.ransac(table, 0.01)
        got = sg.global_counters(0)
    finally:
        sg.close()
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    grid.subdivide([lambda p: len(p) > 64])
    np.random.seed(0)
    grid.map_leaf_points_cuda_ransac()
    assert got == (grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0))
