"""csrc/route.hip with MORE THAN ONE RANK, on the one GPU a test box has.

Real RCCL refuses two ranks of a communicator on one device, so the library's RCCL table (resolved with
dlopen) is pointed at a test-only stand-in (tests/rccl_stub/: shared-memory transport between rank PROCESSES,
OCTL_RCCL_LIBRARY).  Everything of the routing that depends on R > 1 then really runs: the R x (R+2) count
matrix, the send / receive offsets, the grouped transfers, the grow agreement, the collective error exits.
Each rank's shard is checked against voxel_owner_np and a single-process build of the same shard."""

import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawner():
    from tests import conftest

    mp = conftest.rank_spawner()
    if mp is None:
        pytest.skip("no fork server (it must be started before the GPU is touched: run through pytest's conftest)")
    return mp


def _build_stub():
    import subprocess

    so = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")
    if not os.path.exists(so):
        pytest.skip("tests/rccl_stub/librccl_stub.so is not built (make stub / __graft_entry__.build())")
    return so


@pytest.mark.parametrize("world", [2, 3])
def test_route_points_with_several_ranks_on_one_gpu(world):
    from octreelib_amd.distributed import voxel_owner_np
    from octreelib_amd.grid import Grid, GridConfig
    from tests._route_worker import clouds_of, run

    _build_stub()
    mp = _spawner()
    K = 32
    conns, procs = [], []
    for r in range(world):
        parent, child = mp.Pipe()
        p = mp.Process(target=run, args=(r, world, child, K), daemon=True)
        p.start()
        conns.append(parent)
        procs.append(p)
    try:
        # the unique id: rank 0 -> parent -> the others (the comm_broadcast plumbing of ShardedGrid)
        assert conns[0].poll(180), "rank 0 did not start"
        tag, uid = conns[0].recv()
        assert tag == "uid"
        for c in conns[1:]:
            c.send(("uid", uid))
        results = []
        for r, c in enumerate(conns):
            assert c.poll(300), f"rank {r} did not finish (a rank waiting in a collective?)"
            tag, out = c.recv()
            assert tag == "result"
            results.append(out)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for out in results:
        assert out["ok"], out.get("error")
    # ---- expectations, computed here in one process ------------------------------------------------------------
    A, B = zip(*[clouds_of(r, world) for r in range(world)])

    def shard(clouds, base0, q):
        pts, gidx = [], []
        for p, c in enumerate(clouds):   # received in source-rank order, each source's points in its own order
            own = voxel_owner_np(np.floor(c).astype(np.int64), world) == q
            pts.append(c[own])
            gidx.append(base0 + 1_000_000 * p + np.nonzero(own)[0])
        return np.vstack(pts), np.concatenate(gidx)

    def rows_of(grid, pose, gidx, index):
        return [((np.asarray(v.corner_min, dtype=np.float64) + 0.0).tobytes(), np.float64(v.edge_length).tobytes(),
                 tuple(sorted(gidx[[index[p.tobytes()] for p in v.get_points()]].tolist())))
                for v in grid.get_leaf_points(pose)]

    tot1 = np.zeros(3, dtype=np.int64)
    tot2 = np.zeros((2, 3), dtype=np.int64)
    for q, out in enumerate(results):
        p1, g1 = shard(A, 0, q)
        assert out["n1"] == len(p1) and np.array_equal(out["g1"], g1)
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, p1)
        g.subdivide([lambda p: len(p) > K])
        idx1 = {p1[i].tobytes(): i for i in range(len(p1))}
        assert sorted(out["rows1"]) == sorted(rows_of(g, 0, g1, idx1))
        tot1 += np.array([g.n_nodes(0), g.n_leaves(0), g.n_points(0)])
        # the NaN of the last rank reached everybody
        assert out["domain_error"] and "rank %d" % (world - 1) in out["domain_error"]
        # second pose, both poses re-subdivided, RANSAC
        p2, g2 = shard(B, 50_000_000, q)
        assert out["n2"] == len(p2) and np.array_equal(out["g2"], g2)
        g.insert_points(1, p2)
        g.subdivide([lambda p: len(p) > K])
        idx2 = {p2[i].tobytes(): i for i in range(len(p2))}
        assert sorted(out["rows2"][0]) == sorted(rows_of(g, 0, g1, idx1))
        assert sorted(out["rows2"][1]) == sorted(rows_of(g, 1, g2, idx2))
        np.random.seed(3)
        g.map_leaf_points_cuda_ransac(hypotheses_number=256)
        for s in (0, 1):
            tot2[s] += np.array([g.n_nodes(s), g.n_leaves(s), g.n_points(s)])
    for out in results:   # the all-reduced counters are the sums over the shards, on every rank
        assert list(out["counters1"]) == tot1.tolist()
        assert [list(c) for c in out["counters2"]] == tot2.tolist()
    # ... and the shards add up to the unsharded grid
    ref = Grid(GridConfig(voxel_edge_length=1))
    ref.insert_points(0, np.vstack(A))
    ref.subdivide([lambda p: len(p) > K])
    assert tot1.tolist() == [ref.n_nodes(0), ref.n_leaves(0), ref.n_points(0)]
