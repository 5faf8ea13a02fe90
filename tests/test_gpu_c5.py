"""BASELINE config 5 (10^9 points over [0,128)^3, 8 ranks, sharded by top-level voxel) as seen by ONE
rank: the 125 M points it holds after the all-to-all - every voxel it owns (about 262 k of the 2 M, hash
ownership, ~477 points per voxel) - built and fitted on one GPU, checked through size-independent
properties; and the same pipeline at reduced size against the oracle.  (The exchange itself needs 8
GPUs; its device half - destination keys, counts, stable partition - is checked here on a sample of
the full scene for 8 ranks, the rest in tests/test_gpu_sharded.py.)"""

import ctypes as C

import numpy as np
import pytest

from tests.test_gpu_fullsize import _check_points_in_leaf_cubes, _check_structure

pytestmark = pytest.mark.gpu


def test_c5_one_rank_shard_125M_properties():
    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic
    from octreelib_amd._engine import Forest
    from octreelib_amd.distributed import owned_voxel_ids, voxel_indices_np, voxel_owner_np

    dims, n_ranks, K = (128, 128, 128), 8, 64
    ctx = nat.get_context()
    lib = ctx.lib
    # ---- the device half of the exchange on a sample of the whole scene ---------------------------------
    m = 2_000_000
    sample = synthetic.uniform_cloud(m, dims, seed=1000)
    counts = np.zeros(n_ranks, dtype=np.int64)
    out_xyz = np.empty((m, 3))
    out_idx = np.empty(m, dtype=np.int64)
    ctx.check(lib.octl_debug_route_partition(ctx.handle, nat.ptr(sample), m, 0, 1.0, n_ranks, nat.ptr(counts),
                                             nat.ptr(out_xyz), nat.ptr(out_idx)))
    owner = voxel_owner_np(voxel_indices_np(sample, 1.0), n_ranks)
    assert np.array_equal(counts, np.bincount(owner, minlength=n_ranks))
    assert counts.min() > 0.9 * m / n_ranks and counts.max() < 1.1 * m / n_ranks   # the hash balances
    order = np.argsort(owner, kind="stable")
    assert np.array_equal(out_idx, order) and np.array_equal(out_xyz, sample[order])
    del sample, out_xyz, out_idx
    # ---- rank 0's shard: uniform points in the voxels it owns ---------------------------------------------
    vox = owned_voxel_ids(dims, 0, n_ranks)
    assert abs(len(vox) - 128 ** 3 // 8) < 3000
    n = 125_000_000
    f = Forest(0, np.zeros(3), 1.0)
    d_xyz = C.c_void_p()
    ctx.check(lib.octl_dev_alloc(ctx.handle, n * 24, C.byref(d_xyz)))
    chunk = 12_500_000
    first = None
    for c in range(n // chunk):
        pts = np.ascontiguousarray(synthetic.uniform_cloud(chunk, dims, seed=2000 + c, voxels=vox))
        ctx.check(lib.octl_dev_upload(ctx.handle, C.c_void_p(d_xyz.value + c * chunk * 24), nat.ptr(pts), pts.nbytes))
        if first is None:
            first = pts
    f.add_pose_device(d_xyz, n)
    ctx.set_profiling(True)
    f.subdivide(K)
    names = set(ctx.timings())
    ctx.set_profiling(False)
    assert "bucket_build" in names and "bucket_bounds" in names and "level_hist" not in names
    assert f.info.n_voxels == len(vox)          # 125 M points leave none of the ~262 k owned voxels empty
    assert int(f.info.n_levels) >= 1
    nd, blk, perm = _check_structure(f, n, K)
    # the voxel table is exactly the owned voxels, in lexicographic order
    v = f.voxels
    assert np.array_equal((v[:, 0] * dims[1] + v[:, 1]) * dims[2] + v[:, 2], vox)
    # points of the first uploaded chunk sit in leaves whose cube contains them
    xyz_head = f.xyz[:1]  # (forces the download once)
    xyz_ord = f.xyz
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    assert np.array_equal(xyz_ord[inv[:chunk]], first)
    _check_points_in_leaf_cubes(f, xyz_ord)
    inv_of = lambda c: inv[c * chunk : (c + 1) * chunk]
    del xyz_head
    # ---- RANSAC on a sample of blocks with details, then on everything ---------------------------------------
    np.random.seed(0)
    table = np.random.random((1024, 6))
    order = f.order
    assert np.array_equal(np.sort(order), np.arange(len(order)))
    sub = order[:: max(1, len(order) // 40_000)]
    plane, count, index = f.ransac_blocks(sub, table, 0.01, details=True)
    mask = f.device_mask()
    starts, sizes = blk["start"][sub], blk["size"][sub]
    cs = np.concatenate(([0], np.cumsum(mask, dtype=np.int64)))
    assert np.array_equal(cs[starts + sizes] - cs[starts], count)
    assert (count[sizes < 6] == 0).all() and (count <= sizes).all()
    # ---- the oracle at this size -------------------------------------------------------------------------------
    from tests._fullsize import (oracle_check_every_leaf, oracle_check_ransac_blocks, oracle_check_voxels,
                                 pick_blocks_of_every_size)

    def chunks():
        for c in range(n // chunk):
            yield c * chunk, np.ascontiguousarray(synthetic.uniform_cloud(chunk, dims, seed=2000 + c, voxels=vox))

    rng = np.random.default_rng(13)
    # 200 whole voxels (their ~477 points collected from all ten uploaded chunks) rebuilt by the recursive oracle
    nv, nl = oracle_check_voxels(f, chunks(), 200, K, rng)
    assert nv == 200 and nl > 2000
    # >= 2500 blocks of every size through the operator: count, winner, plane bits, mask
    sel = pick_blocks_of_every_size(blk["size"], 40, rng, at_least=2500)
    sz, _ = oracle_check_ransac_blocks(f, sel, table, 0.01, xyz_ord)
    assert len(sz) >= 2500 and len(np.unique(sz)) >= K - 2
    # every leaf and the leaf of every one of the 125 M points against the count-only oracle (the ten chunks
    # as consecutive pieces of the one pose)
    pieces = [p for _, p in chunks()]
    for c, piece in enumerate(pieces):
        assert np.array_equal(xyz_ord[inv_of(c)], piece)   # the stored cloud IS the uploaded one
    del xyz_ord
    f._xyz = None  # (3 GB: the forest's cached host copy is not needed any more)
    oracle_check_every_leaf(f, pieces, K, grid=True, one_slot=True)
    del pieces
    f.ransac_all(10, table, 0.01)
    mask = f.device_mask()
    kept = int(mask.sum())
    f.apply_device_mask()
    assert f.n_ord == kept and 0 < kept < n
    blk2 = f.blocks
    assert int(blk2["size"].sum()) == kept and np.array_equal(blk2["start"][1:], np.cumsum(blk2["size"])[:-1])
    f.close()
    ctx.check(lib.octl_dev_free(ctx.handle, d_xyz))


def test_c5_reduced_size_shard_against_oracle():
    """The same: one rank's shard of an 8-rank scene (16^3 voxels, ~477 points per owned voxel), the
    whole pipeline through the drop-in classes against the oracle."""
    from octreelib_amd import synthetic
    from octreelib_amd.distributed import owned_voxel_ids
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp
    from oracle import ransac_np as rnp
    from tests._util import assert_same_leaves, canon_from_list
    from tests.test_gpu_parity import index_map, views_table

    dims = (16, 16, 16)
    vox = owned_voxel_ids(dims, 0, 8)
    pts = np.unique(synthetic.planar_cloud(len(vox) * 477, dims, seed=1, stream=5, voxels=vox), axis=0)
    np.random.default_rng(3).shuffle(pts)
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    grid.subdivide([lambda p: len(p) > 64])
    og = onp.OGrid(1)
    og.insert_points(0, pts)
    og.subdivide(64)
    index = index_map(pts)
    assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(0), index)),
                       canon_from_list(og.leaf_table(0)))
    assert [grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0)] == [og.n_nodes(0), og.n_leaves(0), og.n_points(0)]
    np.random.seed(0)
    table = np.random.random((1024, 6))
    np.random.seed(0)
    grid.map_leaf_points_cuda_ransac()
    rows = og.leaf_table(0)
    mask = rnp.evaluate(np.vstack([pts[i] for _, _, i in rows]),
                        np.array([len(i) for _, _, i in rows], dtype=np.int32), table, 0.01)
    og.apply_mask(0, mask)
    assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(0), index)),
                       canon_from_list(og.leaf_table(0)))
    assert grid.n_points(0) == og.n_points(0)


def test_eight_rank_partition_of_a_skewed_scene_is_balanced():
    """The device half of the exchange for 8 ranks on a scene that is NOT uniform (synthetic.sparse_scene: terrain
    sheet + one blob at 20 x the density): counts and packed send buffers equal the host mirror's stable partition,
    and the largest shard stays within 25 % of the mean (max / mean points per rank)."""
    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic
    from octreelib_amd.distributed import voxel_indices_np, voxel_owner_np

    ctx = nat.get_context()
    lib = ctx.lib
    m, n_ranks = 3_000_000, 8
    pts = synthetic.sparse_scene(m, (256, 256, 32), seed=7)
    counts = np.zeros(n_ranks, dtype=np.int64)
    out_xyz = np.empty((m, 3))
    out_idx = np.empty(m, dtype=np.int64)
    ctx.check(lib.octl_debug_route_partition(ctx.handle, nat.ptr(pts), m, 0, 1.0, n_ranks, nat.ptr(counts),
                                             nat.ptr(out_xyz), nat.ptr(out_idx)))
    owner = voxel_owner_np(voxel_indices_np(pts, 1.0), n_ranks)
    assert np.array_equal(counts, np.bincount(owner, minlength=n_ranks))
    order = np.argsort(owner, kind="stable")
    assert np.array_equal(out_idx, order) and np.array_equal(out_xyz, pts[order])
    assert counts.max() / counts.mean() <= 1.25, counts
