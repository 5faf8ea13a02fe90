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


def test_route_ahead_on_a_second_context(monkeypatch):
    """bench.py routes step i+1 on a second context (own stream + communicator, second host
    thread) while step i is computed: the cloud routed there, handed over with
    octl_forest_add_pose_routed_from, must build the same forest as a direct insert."""
    import ctypes as C
    import threading

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic

    monkeypatch.setenv("OCTL_ROUTE_SELF_SENDRECV", "1")
    ctx, rctx = nat.get_context(), nat.Context(0)
    lib = ctx.lib
    uid = (C.c_uint8 * nat.UNIQUE_ID_BYTES)()
    rctx.check(lib.octl_comm_unique_id(C.cast(uid, C.c_void_p)))
    rctx.check(lib.octl_comm_init(rctx.handle, 1, 0, C.cast(uid, C.c_void_p)))
    corner = np.zeros(3)
    clouds = [np.ascontiguousarray(synthetic.planar_cloud(60_000, (4, 4, 4), seed=2, stream=s)) for s in range(3)]
    d_xyz = []
    for c in clouds:
        p = C.c_void_p()
        ctx.check(lib.octl_dev_alloc(ctx.handle, c.nbytes, C.byref(p)))
        ctx.check(lib.octl_dev_upload(ctx.handle, p, nat.ptr(c), c.nbytes))
        d_xyz.append(p)
    fh, fref = C.c_void_p(), C.c_void_p()
    ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fh)))
    ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fref)))
    info, iref, slot, n_recv = nat.BuildInfo(), nat.BuildInfo(), C.c_int32(0), C.c_int64(0)
    routed, free, errors = threading.Event(), threading.Event(), []
    free.set()

    def router():
        try:
            for p, c in zip(d_xyz, clouds):
                free.wait()
                free.clear()
                rctx.check(lib.octl_route_points(rctx.handle, p, None, len(c), 0, nat.ptr(corner), 1.0,
                                                 C.byref(n_recv), None))
                routed.set()
        except BaseException as exc:  # pragma: no cover
            errors.append(exc)
            routed.set()

    th = threading.Thread(target=router)
    th.start()
    try:
        for p, c in zip(d_xyz, clouds):
            routed.wait()
            routed.clear()
            assert not errors, errors
            assert n_recv.value == len(c)
            ctx.check(lib.octl_forest_clear(fh))
            ctx.check(lib.octl_forest_add_pose_routed_from(fh, rctx.handle, C.byref(slot)))
            free.set()  # (no wait: the call has swapped buffers with the router, or waited for its copy)
            ctx.check(lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info)))
            ctx.check(lib.octl_forest_clear(fref))
            ctx.check(lib.octl_forest_add_pose_device(fref, p, len(c), C.byref(slot)))
            ctx.check(lib.octl_forest_build(fref, 32, None, 0, 0, 0, C.byref(iref)))
            assert (info.n_points, info.n_voxels, info.n_nodes, info.n_blocks) == (
                iref.n_points, iref.n_voxels, iref.n_nodes, iref.n_blocks)
            a, b = np.empty((len(c), 3)), np.empty((len(c), 3))
            ctx.check(lib.octl_forest_get_points(fh, 0, len(c), nat.ptr(a)))
            ctx.check(lib.octl_forest_get_points(fref, 0, len(c), nat.ptr(b)))
            assert np.array_equal(a, b)
    finally:
        free.set()
        th.join()
        lib.octl_forest_destroy(fh)
        lib.octl_forest_destroy(fref)
        for p in d_xyz:
            lib.octl_dev_free(ctx.handle, p)
        lib.octl_comm_destroy(rctx.handle)
        rctx.close()


def test_routed_cloud_is_handed_over_once(monkeypatch):
    """An empty forest takes the router's receive buffer over (no copy): the routed cloud is then
    gone and a second hand-over is refused; a forest that already holds a pose copies the cloud
    behind it, and the two poses read back as inserted."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic

    monkeypatch.setenv("OCTL_ROUTE_SELF_SENDRECV", "1")
    ctx = nat.Context(0)
    lib = ctx.lib
    uid = (C.c_uint8 * nat.UNIQUE_ID_BYTES)()
    ctx.check(lib.octl_comm_unique_id(C.cast(uid, C.c_void_p)))
    ctx.check(lib.octl_comm_init(ctx.handle, 1, 0, C.cast(uid, C.c_void_p)))
    corner = np.zeros(3)
    clouds = [np.ascontiguousarray(synthetic.planar_cloud(n, (3, 3, 3), seed=6, stream=s)) for s, n in ((0, 20_001), (1, 9_999))]
    fh, slot, n_recv = C.c_void_p(), C.c_int32(0), C.c_int64(0)
    ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fh)))
    dev = []
    try:
        for k, c in enumerate(clouds):
            p = C.c_void_p()
            ctx.check(lib.octl_dev_alloc(ctx.handle, c.nbytes, C.byref(p)))
            ctx.check(lib.octl_dev_upload(ctx.handle, p, nat.ptr(c), c.nbytes))
            dev.append(p)
            ctx.check(lib.octl_route_points(ctx.handle, p, None, len(c), 0, nat.ptr(corner), 1.0,
                                            C.byref(n_recv), None))
            ctx.check(lib.octl_forest_add_pose_routed(fh, C.byref(slot)))
            assert slot.value == k
            if k == 0:   # taken over: nothing left to hand out
                with pytest.raises(ValueError, match="handed to a forest already"):
                    ctx.check(lib.octl_forest_add_pose_routed(fh, C.byref(slot)))
        info, iref, fref = nat.BuildInfo(), nat.BuildInfo(), C.c_void_p()
        ctx.check(lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info)))
        ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fref)))
        try:
            for c in clouds:
                ctx.check(lib.octl_forest_add_pose(fref, nat.ptr(c), len(c), C.byref(slot)))
            ctx.check(lib.octl_forest_build(fref, 32, None, 0, 0, 0, C.byref(iref)))
            n = sum(len(c) for c in clouds)
            assert (info.n_points, info.n_voxels, info.n_nodes, info.n_blocks) == (
                n, iref.n_voxels, iref.n_nodes, iref.n_blocks)
            a, b = np.empty((n, 3)), np.empty((n, 3))
            ctx.check(lib.octl_forest_get_points(fh, 0, n, nat.ptr(a)))
            ctx.check(lib.octl_forest_get_points(fref, 0, n, nat.ptr(b)))
            assert np.array_equal(a, b)
        finally:
            lib.octl_forest_destroy(fref)
    finally:
        lib.octl_forest_destroy(fh)
        for p in dev:
            lib.octl_dev_free(ctx.handle, p)
        lib.octl_comm_destroy(ctx.handle)
        ctx.close()


@pytest.mark.parametrize("n_ranks", [2, 3, 8])
def test_device_partition_for_many_ranks_and_emulated_exchange(n_ranks):
    """The communicator-independent half of the routing (destination keys, counts, stable partition,
    packing) for R > 1 ranks on one GPU, against NumPy; then the all-to-all emulated on the host:
    every virtual rank builds its shard, and the shards add up to the unsharded grid."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic
    from octreelib_amd.distributed import voxel_owner_np
    from octreelib_amd.grid import Grid, GridConfig

    ctx = nat.get_context()
    lib = ctx.lib
    L = 1.0
    clouds = [np.ascontiguousarray(synthetic.planar_cloud(30_000, (6, 5, 4), seed=4, stream=r)) for r in range(n_ranks)]
    clouds[0][:50] -= 3.0   # negative voxel indices too
    send = []
    for r, c in enumerate(clouds):
        counts = np.zeros(n_ranks, dtype=np.int64)
        xyz_out, gidx_out = np.empty_like(c), np.empty(len(c), dtype=np.int64)
        base = 1_000_000 * r
        ctx.check(lib.octl_debug_route_partition(ctx.handle, nat.ptr(c), len(c), base, L, n_ranks,
                                                 nat.ptr(counts), nat.ptr(xyz_out), nat.ptr(gidx_out)))
        owner = voxel_owner_np(np.floor(c / L).astype(np.int64), n_ranks)
        order = np.argsort(owner, kind="stable")
        assert np.array_equal(counts, np.bincount(owner, minlength=n_ranks))
        assert np.array_equal(gidx_out, base + order)           # stable partition by destination
        assert np.array_equal(xyz_out, c[order])
        off = np.concatenate(([0], np.cumsum(counts)))
        send.append([(xyz_out[off[q]:off[q + 1]], gidx_out[off[q]:off[q + 1]]) for q in range(n_ranks)])
    # emulated all-to-all: rank q receives, in source-rank order, what every rank p packed for it
    total = np.zeros(3, dtype=np.int64)
    for q in range(n_ranks):
        recv = np.vstack([send[p][q][0] for p in range(n_ranks)])
        assert (voxel_owner_np(np.floor(recv / L).astype(np.int64), n_ranks) == q).all()
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, recv)
        g.subdivide([lambda p: len(p) > 32])
        total += np.array([g.n_nodes(0), g.n_leaves(0), g.n_points(0)])
    ref = Grid(GridConfig(voxel_edge_length=1))
    ref.insert_points(0, np.vstack(clouds))
    ref.subdivide([lambda p: len(p) > 32])
    assert total.tolist() == [ref.n_nodes(0), ref.n_leaves(0), ref.n_points(0)]
