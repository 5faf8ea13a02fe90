"""
Parity of the HIP path (through the C ABI and the drop-in Python classes) with
  * the golden vectors produced by the reference itself (tests/golden/*.npz),
  * the known answers of the reference's own tests,
  * the oracle (oracle/octree_np.py, oracle/ransac_np.py) on seeded inputs.
Bar: bit-exact leaf tables (corner bits, edge bits) -> original point index sets, leaf list
order, counters; RANSAC inlier counts / winning hypothesis / f32 plane / mask exact.
"""

import numpy as np
import pytest

from tests._util import assert_same_leaves, canon_from_list, golden_canon, load_golden, set_option

pytestmark = pytest.mark.gpu


def index_map(points):
    pts = np.ascontiguousarray(points, dtype=np.float64)
    d = {pts[i].tobytes(): i for i in range(len(pts))}
    assert len(d) == len(pts)
    return d


def views_table(leaves, index):
    out = []
    for v in leaves:
        p = np.ascontiguousarray(v.get_points(), dtype=np.float64)
        out.append((np.asarray(v.corner_min, dtype=np.float64), np.float64(v.edge_length),
                    [index[p[i].tobytes()] for i in range(len(p))]))
    return out


def crit(k):
    return [lambda pts: len(pts) > k]


# ------------------------------------------------------------------------------------------------
# golden vectors from the reference
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2000, 20000])
@pytest.mark.parametrize("k", [8, 32, 256])
def test_octree_uniform_golden(n, k):
    from octreelib_amd.octree import Octree, OctreeConfig

    g = load_golden(f"octree_uniform_{n}.npz")
    pts = g["points"]
    index = index_map(pts)
    oc = Octree(OctreeConfig(), np.array([0.0, 0.0, 0.0]), np.float64(1))
    oc.insert_points(pts)
    assert (oc.get_points() == pts).all()  # insertion order before any subdivide
    oc.subdivide(crit(k))
    assert_same_leaves(canon_from_list(views_table(oc.get_leaf_points(), index)), golden_canon(g, f"k{k}"))
    assert [oc.n_nodes, oc.n_leaves, oc.n_points] == list(g[f"k{k}_counts"])
    all_leaves = oc.get_leaf_points(non_empty=False)
    assert np.array_equal(np.array([v.corner_min for v in all_leaves]), g[f"k{k}_all_corners"])
    assert np.array_equal(np.array([v.edge_length for v in all_leaves]), g[f"k{k}_all_edges"])
    # within a leaf the points keep their insertion order (stable), DFS order overall
    got = oc.get_points()
    assert sorted(map(bytes, got)) == sorted(map(bytes, pts))


def test_grid_L1_mixed_golden():
    from octreelib_amd.grid import Grid, GridConfig

    g = load_golden("grid_L1_mixed.npz")
    pts = g["points"]
    index = index_map(pts)
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(0), index)), golden_canon(g, "pre"))
    assert [grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0)] == list(g["pre_counts"])
    grid.subdivide(crit(16))
    assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(0), index)), golden_canon(g, "k16"))
    assert [grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0)] == list(g["k16_counts"])


def test_grid_L5_two_poses_golden():
    from octreelib_amd.grid import Grid, GridConfig

    g = load_golden("grid_L5_two_poses.npz")
    grid = Grid(GridConfig(voxel_edge_length=5))
    idx = []
    for p in range(2):
        grid.insert_points(p, g[f"points{p}"])
        idx.append(index_map(g[f"points{p}"]))
    grid.subdivide(crit(24))
    for p in range(2):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"p{p}"))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"p{p}_counts"])
    grid.subdivide(crit(6), [1])  # refinement driven by pose 1 only
    for p in range(2):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"r_p{p}"))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"r_p{p}_counts"])


@pytest.mark.parametrize("incremental", [True, False])
def test_grid_late_poses_golden(monkeypatch, incremental):
    """The reference's own leaf tables for poses inserted after a subdivide - through the incremental
    insertion (incremental.hip) and through the re-placement of every stored point."""
    from octreelib_amd.grid import Grid, GridConfig

    if incremental:
        set_option("NO_INCREMENTAL", 0)
    else:
        set_option("NO_INCREMENTAL", 1)
    g = load_golden("grid_late_poses.npz")
    grid = Grid(GridConfig(voxel_edge_length=2))
    idx = [index_map(g[f"points{p}"]) for p in range(5)]

    def check(tag, n):
        for p in range(n):
            assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"{tag}_p{p}"))
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    grid.insert_points(0, g["points0"])
    grid.subdivide(crit(60))
    grid.insert_points(1, g["points1"])
    check("a", 2)
    grid.insert_points(2, g["points2"])
    check("b", 3)
    grid.insert_points(3, g["points3"])
    grid.insert_points(4, g["points4"])
    check("c", 5)
    grid.subdivide(crit(25))
    check("d", 5)


def test_manager_four_poses_golden():
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager

    g = load_golden("manager_four_poses.npz")
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 2.0)
    idx = [index_map(g[f"points{p}"]) for p in range(4)]
    for p in range(3):
        m.insert_points(p, g[f"points{p}"])
    m.subdivide(crit(40), [0, 2])
    m.insert_points(3, g["points3"])  # inherits the scheme
    for p in range(4):
        got = canon_from_list(views_table(m.get_leaf_points(True, p), idx[p]))
        assert_same_leaves(got, golden_canon(g, f"p{p}"))
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == list(g[f"p{p}_counts"])
    m.subdivide(crit(25))
    for p in range(4):
        got = canon_from_list(views_table(m.get_leaf_points(True, p), idx[p]))
        assert_same_leaves(got, golden_canon(g, f"r_p{p}"))
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == list(g[f"r_p{p}_counts"])


@pytest.mark.parametrize("name", ["h64", "h1024", "h32k3"])
def test_ransac_operator_golden_and_oracle(name):
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    g = load_golden(f"ransac_{name}.npz")
    cloud, sizes, hyp, thr = g["cloud"], g["block_sizes"], g["hypotheses"], float(g["threshold"])
    H, k = hyp.shape
    np.random.seed(1000 + H)  # same draw as the generator: the table must come out identical
    op = CudaRansac(threshold=thr, hypotheses_number=H, initial_points_number=k)
    assert np.array_equal(op.random_hypotheses, hyp)
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    o_mask, o_count, o_plane, o_index, tied = rnp.evaluate(cloud, sizes, hyp, thr, details=True)
    # exact against the oracle
    assert np.array_equal(counts, o_count)
    assert np.array_equal(index, o_index)
    assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
    assert np.array_equal(mask, o_mask)
    # against the reference kernel source: the winner among ties is a race upstream, so its mask
    # must be the mask of ONE of the tied planes (checked directly against the HIP results)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for b in range(len(sizes)):
        ref = g["mask"][starts[b] : starts[b + 1]]
        assert int(ref.sum()) == int(counts[b])
        if sizes[b] >= k:
            blk = cloud[starts[b] : starts[b + 1]]
            assert any(np.array_equal(_plane_mask(t, blk, thr), ref) for t in tied[b])
            assert any(np.array_equal(planes[b].view(np.uint32), t.view(np.uint32)) for t in tied[b])


def _plane_mask(plane32, blk, thr):
    """measure_distance(plane, point) < threshold (util.py:22-24) for one f32 plane."""
    p = np.asarray(plane32, dtype=np.float32).astype(np.float64)
    return np.abs(((p[0] * blk[:, 0] + p[1] * blk[:, 1]) + p[2] * blk[:, 2]) + p[3]) < thr


@pytest.mark.parametrize("name", ["h1024", "h256"])
def test_ransac_unique_maximiser_blocks_against_the_reference_plane(name):
    """208 blocks on which one hypothesis attains the maximum: the reference kernel's own best_plane (no tie for
    its race to decide) - HIP plane within 1e-5 on the normal (the same f32 bits), count and mask equal."""
    from octreelib_amd.ransac import CudaRansac

    g = load_golden(f"ransac_unique_{name}.npz")
    cloud, sizes, hyp, thr = g["cloud"], g["block_sizes"], g["hypotheses"], float(g["threshold"])
    H, k = hyp.shape
    np.random.seed(int(g["seed"]))
    op = CudaRansac(threshold=thr, hypotheses_number=H, initial_points_number=k)
    assert np.array_equal(op.random_hypotheses, hyp)
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    assert np.array_equal(counts, g["ref_max_inliers"])
    assert np.max(np.abs(planes[:, :3].astype(np.float64) - g["ref_plane"][:, :3])) <= 1e-5
    assert np.array_equal(planes.view(np.uint32), g["ref_plane"].view(np.uint32))
    assert np.array_equal(mask, g["mask"])


@pytest.mark.parametrize("name", ["h1024", "h256"])
def test_ransac_bench_leaves_against_recorded_reference_planes(name):
    """Blocks cut from the benchmark scene's own leaves (sizes 6..64, a few up to ~305) with the
    reference kernel's shared best_plane / max_inliers_number recorded per block
    (cuda_ransac.py:125-146): HIP best_count == the reference's maximum; the HIP mask is the mask of
    one of the tied planes; where a single plane attains the maximum, the HIP plane is the
    reference's (|delta normal| <= 1e-5, in fact the same f32 bits) and the masks are equal."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    g = load_golden(f"ransac_bench_leaves_{name}.npz")
    cloud, sizes, hyp, thr = g["cloud"], g["block_sizes"], g["hypotheses"], float(g["threshold"])
    ref_plane, ref_max = g["ref_plane"], g["ref_max_inliers"]
    H, k = hyp.shape
    np.random.seed(int(g["seed"]))
    op = CudaRansac(threshold=thr, hypotheses_number=H, initial_points_number=k)
    assert np.array_equal(op.random_hypotheses, hyp)
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    o_mask, o_count, o_plane, o_index, tied = rnp.evaluate(cloud, sizes, hyp, thr, details=True)
    assert np.array_equal(counts, o_count) and np.array_equal(index, o_index)
    assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
    assert np.array_equal(mask, o_mask)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    singles = 0
    for b in range(len(sizes)):
        s, e = starts[b], starts[b + 1]
        ref = g["mask"][s:e]
        if sizes[b] < k:
            assert not mask[s:e].any() and not ref.any()
            continue
        assert int(counts[b]) == int(ref_max[b]), f"block {b}"
        blk = cloud[s:e]
        assert any(np.array_equal(_plane_mask(t, blk, thr), mask[s:e]) for t in tied[b])
        assert any(np.array_equal(ref_plane[b].view(np.uint32), t.view(np.uint32)) for t in tied[b])
        if len(tied[b]) == 1:
            singles += 1
            assert np.max(np.abs(planes[b][:3].astype(np.float64) - ref_plane[b][:3])) <= 1e-5
            assert np.array_equal(planes[b].view(np.uint32), ref_plane[b].view(np.uint32))
            assert np.array_equal(mask[s:e], ref)
    assert singles >= 5  # (most planar leaves are explained by several hypotheses: ties are the rule)


def test_grid_ransac_end_to_end_golden():
    from octreelib_amd.grid import Grid, GridConfig

    g = load_golden("grid_ransac_e2e.npz")
    grid = Grid(GridConfig(voxel_edge_length=5))
    idx = []
    for p in range(2):
        grid.insert_points(p, g[f"points{p}"])
        idx.append(index_map(g[f"points{p}"]))
    np.random.seed(int(g["seed"]))
    grid.map_leaf_points_cuda_ransac(poses_per_batch=10, threshold=0.01, hypotheses_number=256,
                                     initial_points_number=6)
    for p in range(2):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"p{p}"))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"p{p}_counts"])


# ------------------------------------------------------------------------------------------------
# the reference's own tests, run against the drop-in classes
# ------------------------------------------------------------------------------------------------
def test_reference_test_octree():
    from octreelib_amd.octree import Octree, OctreeConfig, OctreeNode

    pc = np.array([[0, 0, 1], [0, 0, 2], [0, 0, 3], [9, 9, 8], [9, 9, 9]], dtype=float)
    cached = []
    node = OctreeNode(np.array([0, 0, 0]), np.float64(10), cached)
    node.insert_points(pc)
    node.subdivide([lambda points: len(points) > 2])
    assert node.n_leaves == 3 and node.n_points == 5
    node.filter([lambda points: len(points) >= 2])
    assert node.n_points == 4
    assert len(cached) == 15
    oc = Octree(OctreeConfig(), np.array([0, 0, 0]), np.float64(10))
    oc.insert_points(pc)
    assert (pc == oc.get_points()).all()
    oc.subdivide([lambda points: len(points) > 2])
    assert oc.n_leaves == 3 and oc.n_points == 5
    oc.filter([lambda points: len(points) >= 2])
    assert oc.n_points == 4


def _multi_pose():
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager

    m = OctreeManager(Octree, OctreeConfig(), np.array([0, 0, 0]), 5)
    p0 = np.array([[0, 0, 1], [0, 0, 2], [0, 0, 3]], dtype=float)
    p1 = np.array([[1, 0, 1], [4, 0, 2], [0, 2, 3]], dtype=float)
    m.insert_points(0, p0)
    m.insert_points(1, p1)
    return m, {0: p0, 1: p1}


def _same(a, b):
    return set(map(str, np.asarray(a).tolist())) == set(map(str, np.asarray(b).tolist()))


def test_reference_test_multi_pose():
    from octreelib_amd.internal import Voxel

    m, clouds = _multi_pose()
    assert _same(m.get_points(0), clouds[0]) and _same(m.get_points(1), clouds[1])
    assert [m.n_nodes(0), m.n_nodes(1), m.n_leaves(0), m.n_leaves(1)] == [1, 1, 1, 1]
    assert [m.n_points(0), m.n_points(1)] == [3, 3]
    m.subdivide([lambda points: len(points) > 2], [0])
    assert [m.n_nodes(0), m.n_nodes(1)] == [9, 9]
    assert [m.n_leaves(0), m.n_leaves(1)] == [2, 3]
    exp0 = [Voxel(np.array([0, 0, 0]), 2.5), Voxel(np.array([0, 0, 2.5]), 2.5)]
    exp1 = [Voxel(np.array([0, 0, 0]), 2.5), Voxel(np.array([0, 0, 2.5]), 2.5), Voxel(np.array([2.5, 0, 0]), 2.5)]
    assert {v.id for v in m.get_leaf_points(pose_number=0)} == {v.id for v in exp0}
    assert {v.id for v in m.get_leaf_points(pose_number=1)} == {v.id for v in exp1}

    m, clouds = _multi_pose()
    m.subdivide([lambda points: len(points) > 1], None)
    assert [m.n_nodes(0), m.n_nodes(1)] == [33, 33]
    assert [m.n_leaves(0), m.n_leaves(1)] == [3, 3]
    exp0 = [Voxel(np.array([0, 0, 0.625]), 0.625), Voxel(np.array([0, 0, 1.25]), 1.25), Voxel(np.array([0, 0, 2.5]), 1.25)]
    exp1 = [Voxel(np.array([0.625, 0, 0.625]), 0.625), Voxel(np.array([0, 1.25, 2.5]), 1.25), Voxel(np.array([2.5, 0, 0]), 2.5)]
    assert {v.id for v in m.get_leaf_points(pose_number=0)} == {v.id for v in exp0}
    assert {v.id for v in m.get_leaf_points(pose_number=1)} == {v.id for v in exp1}

    m, clouds = _multi_pose()
    m.map_leaf_points(lambda points: points[0].reshape((1, 3)), [0])
    assert [m.n_points(0), m.n_points(1)] == [1, 3]
    m, clouds = _multi_pose()
    m.subdivide([lambda points: len(points) > 2], [0])
    m.filter([lambda points: False], [0])
    m.filter([lambda points: True], [1])
    assert [m.n_points(0), m.n_points(1)] == [0, 3]


def _grid():
    from octreelib_amd.grid import Grid, GridConfig

    grid = Grid(GridConfig(voxel_edge_length=5))
    p0 = np.array([[0, 0, 1], [0, 0, 2], [0, 0, 3], [9, 9, 8], [9, 9, 9]], dtype=float)
    p1 = np.array([[1, 0, 1], [4, 0, 2], [0, 2, 3], [5, 9, 9], [9, 3, 8]], dtype=float)
    grid.insert_points(0, p0)
    grid.insert_points(1, p1)
    return grid, [p0, p1]


def test_reference_test_grid():
    grid, pp = _grid()
    assert [grid.n_leaves(0), grid.n_leaves(1)] == [2, 3]
    assert [grid.n_points(0), grid.n_points(1)] == [5, 5]
    assert [grid.n_nodes(0), grid.n_nodes(1)] == [2, 3]
    assert _same(grid.get_points(0), pp[0]) and _same(grid.get_points(1), pp[1])
    l0, l1 = grid.get_leaf_points(0), grid.get_leaf_points(1)
    assert len({l0[0].id, l0[1].id, l1[0].id, l1[1].id, l1[2].id}) == 3
    assert {v.id for v in l0}.issubset({v.id for v in l1})
    assert _same(l0[0].get_points(), pp[0][:3]) and _same(l0[1].get_points(), pp[0][3:])
    assert _same(l1[0].get_points(), pp[1][:3]) and _same(l1[1].get_points(), pp[1][4:])
    assert _same(l1[2].get_points(), pp[1][3:4])
    grid.subdivide([lambda points: len(points) > 2])
    assert [grid.n_leaves(0), grid.n_leaves(1)] == [4, 5]
    assert [grid.n_points(0), grid.n_points(1)] == [5, 5]
    assert [grid.n_nodes(0), grid.n_nodes(1)] == [26, 27]
    assert _same(grid.get_points(0), pp[0]) and _same(grid.get_points(1), pp[1])
    grid, pp = _grid()
    grid.subdivide([lambda points: len(points) > 3])
    assert [grid.n_leaves(0), grid.n_leaves(1)] == [3, 5]
    grid, pp = _grid()
    assert grid.n_points(0) > grid.n_leaves(0)
    grid.map_leaf_points(lambda cloud: [cloud[0]])
    assert grid.n_points(0) == grid.n_leaves(0) and grid.n_points(1) == grid.n_leaves(1)
    with pytest.raises(ValueError, match="Cannot insert points to existing pose 0"):
        grid.insert_points(0, np.zeros((1, 3)))


def test_reference_test_cuda_ransac_smoke_and_errors():
    from octreelib_amd.grid import Grid, GridConfig

    def planar(n, coef, corner, edge, sigma):
        vp = np.random.rand(n, 3) * np.array([edge - 6 * sigma] * 3) + corner + 3 * sigma
        z = (-coef[0] * vp[:, 0] - coef[1] * vp[:, 1] - coef[3]) / coef[2] + np.random.normal(0, sigma, (n,))
        return np.column_stack((vp[:, :2], z))

    np.random.seed(3)
    grid = Grid(GridConfig(voxel_edge_length=5))
    grid.insert_points(0, planar(10, (1, 2, 3, 0.5), np.array([0, 0, 0]), 5, 0.1))
    grid.insert_points(1, planar(10, (-1, 2, 3, 0.5), np.array([0, 0, 0]), 5, 0.1))
    grid.map_leaf_points_cuda_ransac()
    with pytest.raises(ValueError, match="Threshold must be positive"):
        grid.map_leaf_points_cuda_ransac(threshold=0)
    with pytest.raises(ValueError, match="Number of RANSAC hypotheses must be positive"):
        grid.map_leaf_points_cuda_ransac(hypotheses_number=0)
    with pytest.raises(ValueError, match="must be <= 1024 because of the CUDA thread limit"):
        grid.map_leaf_points_cuda_ransac(hypotheses_number=1025)


# ------------------------------------------------------------------------------------------------
# seeded inputs against the oracle
# ------------------------------------------------------------------------------------------------
def _oracle_pose_table(og, pose):
    return canon_from_list(og.leaf_table(pose))


@pytest.mark.parametrize("n,extent,k", [(60_000, 6.0, 64), (200_000, 8.0, 32)])
def test_grid_uniform_vs_oracle(n, extent, k):
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    pts = np.random.default_rng(n).random((n, 3)) * extent - extent / 3
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    grid.subdivide(crit(k))
    og = onp.OGrid(1)
    og.insert_points(0, pts)
    og.subdivide(k)
    index = index_map(pts)
    assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(0), index)), _oracle_pose_table(og, 0))
    assert [grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0)] == [og.n_nodes(0), og.n_leaves(0), og.n_points(0)]
    # within-leaf order is the insertion order (stable), exactly like the oracle
    f = grid._forest
    blk, perm = f.blocks, f.perm
    for s, z in zip(blk["start"][:200], blk["size"][:200]):
        assert np.all(np.diff(perm[s : s + z]) > 0)


def test_multi_pose_grid_vs_oracle_with_history():
    """3 poses, scheme from a subset, a late pose, a refinement: leaf LIST ORDER depends on the
    history of splits (cached-leaf list), reproduced on the device."""
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(5)
    poses = [rng.random((n, 3)) * 4.0 for n in (9000, 7000, 8000, 5000)]
    grid, og = Grid(GridConfig(voxel_edge_length=2)), onp.OGrid(2)
    for p in range(3):
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])
    grid.subdivide(crit(200), [0, 2])
    og.subdivide(200, [0, 2])
    grid.insert_points(3, poses[3])
    og.insert_points(3, poses[3])
    idx = [index_map(p) for p in poses]
    for p in range(4):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), _oracle_pose_table(og, p))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]
    grid.subdivide(crit(60))
    og.subdivide(60)
    for p in range(4):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), _oracle_pose_table(og, p))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]


def test_deep_clusters_beyond_21_levels_vs_oracle():
    from octreelib_amd.octree import Octree, OctreeConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(9)
    base = rng.random((300, 3))
    clusters = np.vstack([c + rng.random((30, 3)) * 2e-9 for c in rng.random((4, 3)) * 0.8 + 0.1])
    pts = np.unique(np.vstack([base, clusters]), axis=0)
    rng.shuffle(pts)
    oc = Octree(OctreeConfig(), np.array([0.0, 0.0, 0.0]), np.float64(1))
    oc.insert_points(pts)
    oc.subdivide(crit(3))
    t = onp.OTree(np.array([0.0, 0.0, 0.0]), np.float64(1))
    t.insert_points(pts)
    t.subdivide(3)
    index = index_map(pts)
    got = canon_from_list(views_table(oc.get_leaf_points(), index))
    assert_same_leaves(got, canon_from_list(onp.tree_leaf_table(t)))
    assert oc._forest.info.max_depth > 21
    assert [oc.n_nodes, oc.n_leaves, oc.n_points] == [t.n_nodes, t.n_leaves, t.n_points]


def _planar_cloud(rng, n_vox, per_voxel):
    parts = []
    for c in np.argwhere(np.ones((n_vox, n_vox, n_vox))):
        a, b = rng.uniform(-0.4, 0.4, 2)
        n_in = int(per_voxel * 0.8)
        xy = rng.random((n_in, 2))
        z = 0.5 + a * (xy[:, 0] - 0.5) + b * (xy[:, 1] - 0.5) + rng.normal(0, 0.005, n_in)
        inl = np.column_stack([xy, np.clip(z, 0.001, 0.999)])
        out = rng.random((per_voxel - n_in, 3))
        parts.append(np.vstack([inl, out]) + c)
    pts = np.vstack(parts)
    rng.shuffle(pts)
    return pts


def test_grid_ransac_pipeline_vs_oracle():
    """insert + subdivide + RANSAC + apply_mask on two poses against the oracle end to end
    (oracle leaf tables in the same stable order -> oracle kernel restatement -> masks)."""
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(17)
    poses = [_planar_cloud(rng, 3, 300), _planar_cloud(rng, 3, 200)]
    grid, og = Grid(GridConfig(voxel_edge_length=1)), onp.OGrid(1)
    for p, pts in enumerate(poses):
        grid.insert_points(p, pts)
        og.insert_points(p, pts)
    grid.subdivide(crit(64))
    og.subdivide(64)
    np.random.seed(0)
    table = np.random.random((1024, 6))
    np.random.seed(0)
    grid.map_leaf_points_cuda_ransac(poses_per_batch=10, threshold=0.01, hypotheses_number=1024,
                                     initial_points_number=6)
    # oracle: one batch holding both poses (grid.py:149-191)
    clouds, sizes = [], []
    for p, pts in enumerate(poses):
        for _, _, idx in og.leaf_table(p):
            clouds.append(pts[idx])
            sizes.append(len(idx))
    mask = rnp.evaluate(np.vstack(clouds), np.array(sizes, dtype=np.int32), table, 0.01)
    off = 0
    for p, pts in enumerate(poses):
        n = og.n_points(p)
        og.apply_mask(p, mask[off : off + n])
        off += n
    for p, pts in enumerate(poses):
        index = index_map(pts)
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)), _oracle_pose_table(og, p))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]
    # a second subdivide works on the surviving points only
    grid.subdivide(crit(20))
    og.subdivide(20)
    for p, pts in enumerate(poses):
        index = index_map(pts)
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)), _oracle_pose_table(og, p))


def test_ransac_operator_random_blocks_vs_oracle():
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(23)
    sizes = rng.integers(0, 90, 400).astype(np.int32)
    sizes[::37] = 700  # a few large leaves
    cloud = rng.random((int(sizes.sum()), 3)) * 32.0
    # make most blocks roughly planar so that the counts are informative
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for b in range(len(sizes)):
        s, e = starts[b], starts[b + 1]
        if e - s >= 6 and b % 5:
            cloud[s:e, 2] = 3.0 + 0.2 * cloud[s:e, 0] - 0.1 * cloud[s:e, 1] + rng.normal(0, 0.006, e - s)
    for H, k in ((1024, 6), (100, 6), (64, 4), (1024, 5), (300, 9)):
        np.random.seed(H)
        op = CudaRansac(threshold=0.01, hypotheses_number=H, initial_points_number=k)
        mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
        o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.01, details=True)
        assert np.array_equal(counts, o_count)
        assert np.array_equal(index, o_index)
        assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
        assert np.array_equal(mask, o_mask)


@pytest.mark.parametrize("H,k", [(1, 6), (2, 3), (65, 6), (255, 2), (257, 1), (1000, 16), (1024, 7), (64, 16), (1024, 3), (513, 4), (1024, 5),
                                 (1024, 17), (200, 33), (64, 100)])
def test_ransac_operator_boundary_shapes_vs_oracle(H, k):
    """Block sizes around the LDS-staged / global-memory split (255 | 256 points), around k, empty
    blocks; hypothesis counts around the 64 / 256 / 1024 lane mappings; k in registers (<= 16) and
    streamed (any k, as the reference allows)."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(100 * H + k)
    sizes = np.array([255, 256, 257, 254, 0, k, max(k - 1, 0), k + 1, 1, 300, 64, 63, 65, 2, 0, 1023, 17],
                     dtype=np.int32)
    sizes = np.concatenate([sizes, rng.integers(0, 40, 60).astype(np.int32)])
    rng.shuffle(sizes)
    n = int(sizes.sum())
    cloud = rng.random((n, 3)) * 4.0
    cloud[:, 2] = 0.5 * cloud[:, 0] - 0.25 * cloud[:, 1] + rng.normal(0, 0.008, n)
    np.random.seed(H + k)
    op = CudaRansac(threshold=0.01, hypotheses_number=H, initial_points_number=k)
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.01, details=True)
    assert np.array_equal(counts, o_count)
    assert np.array_equal(index, o_index)
    assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
    assert np.array_equal(mask, o_mask)


@pytest.mark.parametrize("H,k", [(1024, 6), (1000, 6), (1024, 3), (600, 5), (1024, 4), (1024, 9)])
def test_ransac_blocks_on_both_sides_of_the_128_point_split_vs_oracle(H, k):
    """With more than 256 hypotheses the size-sorted block list is split on the device: blocks under 128 points go to
    workgroups of 128 lanes x 8 hypotheses (pass 2 in batches of three groups), blocks of 128 ... 255 points to
    256 lanes x 4, larger ones to the tiled kernel.  Sizes on and around every boundary, many blocks of each so that
    every workgroup walks across size changes, exactly planar blocks (the early exit) among them; batches where one
    of the parts is empty."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(1000 * H + k)
    edge = np.array([126, 127, 128, 129, 130, 254, 255, 256, 257, k, k + 1, 64, 65, 96, 191, 192], dtype=np.int32)
    for sizes in (np.concatenate([np.repeat(edge, 9), rng.integers(1, 140, 300).astype(np.int32)]),
                  rng.integers(128, 256, 60).astype(np.int32),        # nothing for the 128-lane instance
                  rng.integers(k, 128, 400).astype(np.int32)):        # nothing for the 256-lane instance
        sizes = sizes.copy()
        rng.shuffle(sizes)
        n = int(sizes.sum())
        cloud = rng.random((n, 3)) * 3.0
        cloud[:, 2] = 0.4 * cloud[:, 0] - 0.2 * cloud[:, 1] + rng.normal(0, 0.008, n)
        starts = np.concatenate(([0], np.cumsum(sizes)))
        for b in range(0, len(sizes), 5):   # exactly planar: some hypothesis of pass 1 holds every point
            s_, e_ = starts[b], starts[b + 1]
            cloud[s_:e_, 2] = 0.5 * cloud[s_:e_, 0] + 0.25 * cloud[s_:e_, 1]
        np.random.seed(H + k)
        op = CudaRansac(threshold=0.01, hypotheses_number=H, initial_points_number=k)
        o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.01, details=True)
        # (RANSAC_WAVES: the launch policy's choice for a launch of this many blocks, then every block on the instance
        #  its size class names / under 128 points on two waves / under 256 on four - ransac.hip: ransac_launch)
        for waves in (0, 1, 2, 4):
            set_option("RANSAC_WAVES", waves)
            mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
            assert np.array_equal(counts, o_count), waves
            assert np.array_equal(index, o_index), waves
            assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32)), waves
            assert np.array_equal(mask, o_mask), waves
        set_option("RANSAC_WAVES", 0)


@pytest.mark.parametrize("H", [1024, 700, 257, 513])
def test_ransac_early_exit_keeps_the_lowest_index_winner(H):
    """The kernel skips the hypotheses H >= 256 of a wavefront once one of its first hypotheses
    holds ALL points of the block (nothing later can beat it; ties go to the lower index).  Many
    small blocks that are fully explained by some hypotheses but not by all: winner index, count,
    plane and mask must still be the reference's, wherever the first full hypothesis sits."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(H)
    sizes = rng.integers(6, 30, 3000).astype(np.int32)
    n = int(sizes.sum())
    cloud = rng.random((n, 3)) * 0.3
    cloud[:, 2] = 0.2 * cloud[:, 0] - 0.1 * cloud[:, 1] + rng.normal(0, 0.0045, n)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for b in range(0, len(sizes), 7):       # exactly planar blocks: every proper hypothesis is full
        s_, e_ = starts[b], starts[b + 1]
        cloud[s_:e_, 2] = 0.25 * cloud[s_:e_, 0] + 0.125 * cloud[s_:e_, 1]
    for b in range(3, len(sizes), 11):      # one far outlier: no hypothesis is full
        cloud[starts[b], 2] += 0.5
    np.random.seed(H)
    op = CudaRansac(threshold=0.01, hypotheses_number=H, initial_points_number=6)
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.01, details=True)
    full = o_count == sizes
    assert 0.1 < full.mean() < 0.95          # the exit is exercised and so is the full evaluation
    for waves in (0, 1, 2, 4):               # (one, two, four waves per block: ransac_launch's policy and each form forced)
        set_option("RANSAC_WAVES", waves)
        mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
        assert np.array_equal(counts, o_count), waves
        assert np.array_equal(index, o_index), waves
        assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32)), waves
        assert np.array_equal(mask, o_mask), waves
    set_option("RANSAC_WAVES", 0)


@pytest.mark.parametrize("H", [1024, 256, 64])
def test_ransac_draws_that_round_up_to_the_next_point_vs_oracle(H):
    """int32(R*n + start) (cuda_ransac.py:103-107): draws with R*n just below an integer round UP once
    the block's start is added, also past the block's end (first point of the next block).  The
    kernel caches sample positions per block size and must take its exact path for those draws."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(77)
    sizes = rng.choice(np.array([37, 37, 37, 64, 12, 6, 100], dtype=np.int32), 600)
    n = int(sizes.sum())
    cloud = rng.random((n, 3)) * 8.0
    cloud[:, 2] = 1.0 + 0.3 * cloud[:, 0] + rng.normal(0, 0.01, n)
    op = CudaRansac(threshold=0.02, hypotheses_number=H, initial_points_number=6)
    table = rng.random((H, 6))
    pick = rng.random((H, 6))
    j = rng.integers(1, 38, (H, 6))
    table = np.where(pick < 0.10, np.nextafter(j / 37.0, 0.0), table)      # just below j/37
    table = np.where((pick >= 0.10) & (pick < 0.15), np.nextafter(1.0, 0.0), table)  # spills past the block
    table = np.where((pick >= 0.15) & (pick < 0.18), j / 64.0 * (1 - 2.0 ** -40), table)
    op._hypotheses = np.ascontiguousarray(table)  # the table is the operator's (cuda_ransac.py:39-41)
    mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.02, details=True)
    assert np.array_equal(counts, o_count)
    assert np.array_equal(index, o_index)
    assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
    assert np.array_equal(mask, o_mask)


@pytest.mark.parametrize(
    "offset,scale,thr",
    [
        (0.0, 1.0, 1.0 / 64),        # lattice: many distances EXACTLY equal to the threshold (strict <)
        (4096.0, 1.0, 1.0 / 64),     # the same far from the origin (block-local screening coordinates)
        (1.0e6, 3.0, 3.0 / 64),      # large offsets, f64 rounding of the offset matters
        (0.0, 1.0e-12, 1.0e-12 / 64),  # tiny scene
        (0.0, 1.0e9, 1.0e9 / 64),    # huge scene
        (0.0, 1.0, 1.0e-30),         # thresholds outside the screened range fall back to exact f64
        (0.0, 1.0, 1.0e30),
    ],
)
def test_ransac_distances_on_the_threshold_vs_oracle(offset, scale, thr):
    """The scoring loop screens point-plane distances in f32 and recounts what it cannot decide:
    the counts must be the exact f64 counts also when many distances sit ON the threshold."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(5)
    sizes = rng.integers(6, 120, 150).astype(np.int32)
    sizes[::29] = [300, 256, 777, 1500, 260, 513][: len(sizes[::29])]  # through the chunked global-memory path
    n = int(sizes.sum())
    # three lattice layers: planes through one layer see the others at exactly 1/64 and 2/64
    cloud = np.empty((n, 3))
    cloud[:, 0] = rng.integers(0, 64, n) / 64.0
    cloud[:, 1] = rng.integers(0, 64, n) / 64.0
    cloud[:, 2] = rng.integers(0, 3, n) / 64.0
    # distinct points inside a block are not required by the operator; a few tilted blocks
    starts = np.concatenate(([0], np.cumsum(sizes)))
    for b in range(0, len(sizes), 4):
        s, e = starts[b], starts[b + 1]
        cloud[s:e, 2] += cloud[s:e, 0] / 2 + rng.integers(0, 2, e - s) / 64.0
    cloud = cloud * scale + offset
    for H in (1024, 200, 64):
        np.random.seed(H + 1)
        op = CudaRansac(threshold=thr, hypotheses_number=H, initial_points_number=6)
        mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
        o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, thr, details=True)
        assert np.array_equal(counts, o_count)
        assert np.array_equal(index, o_index)
        assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
        assert np.array_equal(mask, o_mask)


# ------------------------------------------------------------------------------------------------
# edges of the parity domain
# ------------------------------------------------------------------------------------------------
def test_duplicates_raise_recursion_error():
    from octreelib_amd.octree import Octree, OctreeConfig

    oc = Octree(OctreeConfig(), np.array([0.0, 0.0, 0.0]), np.float64(1))
    oc.insert_points(np.tile(np.array([[0.3, 0.3, 0.3]]), (5, 1)))
    with pytest.raises(RecursionError):
        oc.subdivide(crit(2))


def test_point_outside_cube_only_fails_when_its_node_splits():
    from octreelib_amd.octree import Octree, OctreeConfig

    pts = np.array([[0.1, 0.1, 0.1], [0.2, 0.2, 0.2], [1.5, 0.2, 0.2]])
    oc = Octree(OctreeConfig(), np.array([0.0, 0.0, 0.0]), np.float64(1))
    oc.insert_points(pts)
    oc.subdivide(crit(5))  # no split: the stray point lives in the root, as upstream
    assert oc.n_points == 3 and oc.n_leaves == 1
    oc2 = Octree(OctreeConfig(), np.array([0.0, 0.0, 0.0]), np.float64(1))
    oc2.insert_points(pts)
    with pytest.raises((IndexError, ValueError)):
        oc2.subdivide(crit(1))


def test_tiny_negative_coordinate_is_the_reference_index_error():
    from octreelib_amd.grid import Grid, GridConfig

    pts = np.array([[-5e-324, 0.5, 0.5], [-0.5, 0.5, 0.5], [-0.25, 0.25, 0.5]])
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    assert grid.n_leaves(0) == 1  # all three in voxel (-1, 0, 0)
    with pytest.raises((IndexError, ValueError)):
        grid.subdivide(crit(1))


def test_non_finite_point_is_refused():
    from octreelib_amd.grid import Grid, GridConfig

    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, np.array([[0.5, np.nan, 0.5]]))
    with pytest.raises((ValueError, IndexError)):
        grid.n_leaves(0)


def test_empty_inputs():
    from octreelib_amd.grid import Grid, GridConfig
    from octreelib_amd.octree import Octree, OctreeConfig

    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, np.empty((0, 3)))
    assert [grid.n_leaves(0), grid.n_points(0), grid.n_nodes(0)] == [0, 0, 0]
    grid.subdivide(crit(4))
    assert grid.get_leaf_points(0) == []
    oc = Octree(OctreeConfig(), np.array([0.0, 0.0, 0.0]), np.float64(1))
    assert [oc.n_nodes, oc.n_leaves, oc.n_points] == [1, 0, 0]
    oc.subdivide(crit(4))
    assert [oc.n_nodes, oc.n_leaves, oc.n_points] == [1, 0, 0]


# ------------------------------------------------------------------------------------------------
# arbitrary host callables as criteria (octree.py:26), evaluated level by level by the host
# ------------------------------------------------------------------------------------------------
def test_arbitrary_callable_criteria_vs_oracle():
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(77)
    poses = [rng.random((4000, 3)) * 3.0, rng.random((3000, 3)) * 3.0]

    def spread(points):  # not a count criterion: split while the cloud is both large and wide
        return len(points) > 40 and float(points.max(axis=0).max() - points.min(axis=0).min()) > 0.3

    grid, og = Grid(GridConfig(voxel_edge_length=1)), onp.OGrid(1)
    for p, pts in enumerate(poses):
        grid.insert_points(p, pts)
        og.insert_points(p, pts)
    grid.subdivide([spread], [0])
    og.subdivide([spread], [0])
    for p, pts in enumerate(poses):
        index = index_map(pts)
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)), _oracle_pose_table(og, p))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]
    # a count criterion afterwards refines on the device and inherits the epochs
    grid.subdivide(crit(15))
    og.subdivide(15)
    for p, pts in enumerate(poses):
        index = index_map(pts)
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)), _oracle_pose_table(og, p))


# ------------------------------------------------------------------------------------------------
# batching, pose numbering and the per-pose mask plumbing of map_leaf_points_cuda_ransac
# ------------------------------------------------------------------------------------------------
def _oracle_grid_ransac(og, poses, pose_numbers, table, thr, poses_per_batch):
    """grid.py:149-215 on the oracle: batches of consecutive pose numbers, one evaluate() each."""
    from oracle import ransac_np as rnp

    n = len(pose_numbers)
    for i in range(0, n, poses_per_batch):
        batch = list(range(i, min(i + poses_per_batch, n)))
        clouds, sizes = [], []
        for p in batch:
            for _, _, idx in og.leaf_table(p):
                clouds.append(poses[p][idx])
                sizes.append(len(idx))
        mask = rnp.evaluate(np.vstack(clouds), np.array(sizes, dtype=np.int32), table, thr)
        off = 0
        for p in batch:
            m = og.n_points(p)
            og.apply_mask(p, mask[off : off + m])
            off += m


@pytest.mark.parametrize("poses_per_batch", [1, 2, 10])
def test_ransac_batches_vs_oracle(poses_per_batch):
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(31)
    poses = {p: _planar_cloud(rng, 2, 150 + 40 * p) for p in range(3)}
    grid, og = Grid(GridConfig(voxel_edge_length=1)), onp.OGrid(1)
    for p in range(3):
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])
    grid.subdivide(crit(48))
    og.subdivide(48)
    np.random.seed(5)
    table = np.random.random((512, 6))
    np.random.seed(5)
    grid.map_leaf_points_cuda_ransac(poses_per_batch=poses_per_batch, threshold=0.01,
                                     hypotheses_number=512, initial_points_number=6)
    _oracle_grid_ransac(og, poses, [0, 1, 2], table, 0.01, poses_per_batch)
    for p in range(3):
        index = index_map(poses[p])
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)), _oracle_pose_table(og, p))
        assert [grid.n_leaves(p), grid.n_points(p)] == [og.n_leaves(p), og.n_points(p)]


def test_ransac_poses_inserted_out_of_numeric_order():
    """Pose numbers 1, 0: the reference batches by pose NUMBER (grid.py:149-157); the device order
    is by insertion slot, so this goes through the explicit block-order entry point."""
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(32)
    poses = {0: _planar_cloud(rng, 2, 120), 1: _planar_cloud(rng, 2, 180)}
    grid, og = Grid(GridConfig(voxel_edge_length=1)), onp.OGrid(1)
    for p in (1, 0):
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])
    grid.subdivide(crit(40))
    og.subdivide(40)
    np.random.seed(6)
    table = np.random.random((256, 6))
    np.random.seed(6)
    grid.map_leaf_points_cuda_ransac(poses_per_batch=10, threshold=0.01, hypotheses_number=256)
    _oracle_grid_ransac(og, poses, [0, 1], table, 0.01, 10)
    for p in (0, 1):
        index = index_map(poses[p])
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)), _oracle_pose_table(og, p))


def test_manager_apply_mask_filter_and_map_per_pose():
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp

    rng = np.random.default_rng(33)
    poses = [rng.random((600, 3)), rng.random((500, 3))]
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(2):
        m.insert_points(p, poses[p])
        om.insert_points(p, poses[p])
    m.subdivide(crit(30))
    om.subdivide(30)
    # apply_mask: the mask runs over the pose's non-empty leaves in cached-leaf order
    mask = rng.random(600) < 0.7
    m.apply_mask(mask, 0)
    om.octrees[0].apply_mask(mask)
    for p in range(2):
        index = index_map(poses[p])
        got = canon_from_list(views_table(m.get_leaf_points(True, p), index))
        assert_same_leaves(got, canon_from_list(onp.tree_leaf_table(om.octrees[p])))
    # filter on one pose: leaves with fewer than 12 points are emptied (octree.py:102-112)
    m.filter([lambda pts: len(pts) >= 12], [1])
    for v in om.octrees[1].leaves():
        if len(v.idx) < 12:
            v.idx = np.empty(0, dtype=np.int64)
    # map_leaf_points on the other pose: keep the first point of every leaf (octree.py:114-123)
    m.map_leaf_points(lambda pts: pts[:1], [0])
    for v in om.octrees[0].leaves():
        v.idx = v.idx[:1]
    for p in range(2):
        index = index_map(poses[p])
        got = canon_from_list(views_table(m.get_leaf_points(True, p), index))
        assert_same_leaves(got, canon_from_list(onp.tree_leaf_table(om.octrees[p])))
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == [om.n_nodes(p), om.n_leaves(p), om.n_points(p)]


def test_grid_filter_count_criteria_on_device_vs_oracle_and_host_path():
    """Grid.filter with point-count criteria (octree.py:102-112 through grid.py:260-267) runs as a device
    kernel over the block table + compaction - no download of the cloud; the result equals the oracle's
    and the host path's (the same predicate hidden from the recogniser)."""
    from octreelib_amd import _native as nat
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(77)
    poses = [rng.random((3000, 3)) * 3.0, rng.random((2000, 3)) * 3.0 - 1.0]
    ctx = nat.get_context()

    def make():
        g = Grid(GridConfig(voxel_edge_length=1))
        for p, c in enumerate(poses):
            g.insert_points(p, c)
        g.subdivide(crit(20))
        return g

    for criteria, lo, hi in (([lambda pts: len(pts) >= 5], 5, None), ([lambda pts: len(pts) < 9], None, 8),
                             ([lambda pts: len(pts) > 2, lambda pts: 12 >= len(pts)], 3, 12)):
        g = make()
        ctx.set_profiling(True)
        g.filter(criteria)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        assert "filter" in names                       # the device kernel ran
        assert g._forest._xyz is None                  # ... and nothing was downloaded for it
        og = onp.OGrid(1)
        for p, c in enumerate(poses):
            og.insert_points(p, c)
        og.subdivide(20)
        for key, m in og.managers.items():
            for t in m.octrees.values():
                for v in t.leaves():
                    n = len(v.idx)
                    if (lo is not None and n < lo) or (hi is not None and n > hi):
                        v.idx = np.empty(0, dtype=np.int64)
        h = make()
        opaque = [(lambda f: (lambda pts: bool(f(pts)) and True))(f) for f in criteria]  # not recognisable
        h.filter(opaque)
        for p in range(2):
            index = index_map(poses[p])
            got = canon_from_list(views_table(g.get_leaf_points(p), index))
            assert_same_leaves(got, _oracle_pose_table(og, p))
            assert_same_leaves(canon_from_list(views_table(h.get_leaf_points(p), index)), got)
            assert [g.n_leaves(p), g.n_points(p)] == [og.n_leaves(p), og.n_points(p)]


@pytest.mark.parametrize("plug", ["manager", "octree", "both"])
def test_grid_serves_the_plug_seam_with_user_subclasses(plug):
    """The reference's plug seam (grid_base.py:66-87, grid.py:100-106): GridConfig.octree_manager_type /
    octree_type name the classes the grid instantiates per top-level voxel / per pose.  With a caller's own
    subclasses the grid is served on the host: the instances exist, their (overridden) methods are called, and the
    results - here through insert, subdivide from a pose subset, a late pose, filter, RANSAC - equal the device
    path's and the oracle's.  The reference's own seam test (test_grid.py:148-182: TypeError for non-subclasses)
    is in tests/test_cpu_abi.py."""
    from octreelib_amd.grid import Grid, GridConfig
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp

    calls = {"manager_init": 0, "manager_insert": 0, "octree_init": 0, "octree_subdivide_as": 0}

    class MyManager(OctreeManager):
        def __init__(self, *a, **k):
            calls["manager_init"] += 1
            super().__init__(*a, **k)

        def insert_points(self, pose_number, points):
            calls["manager_insert"] += 1
            return super().insert_points(pose_number, points)

    class MyOctree(Octree):
        def __init__(self, *a, **k):
            calls["octree_init"] += 1
            super().__init__(*a, **k)

        def subdivide_as(self, other):
            calls["octree_subdivide_as"] += 1
            return super().subdivide_as(other)

    cfg = {"manager": dict(octree_manager_type=MyManager), "octree": dict(octree_type=MyOctree),
           "both": dict(octree_manager_type=MyManager, octree_type=MyOctree)}[plug]
    rng = np.random.default_rng(17)
    poses = {p: np.unique(rng.uniform(-2.0, 2.0, (900, 3)), axis=0) for p in range(3)}
    idx = {p: index_map(c) for p, c in poses.items()}
    g = Grid(GridConfig(voxel_edge_length=2, octree_config=OctreeConfig(), **cfg))
    d = Grid(GridConfig(voxel_edge_length=2))
    og = onp.OGrid(2)

    def same(ps):
        for p in ps:
            got = canon_from_list(views_table(g.get_leaf_points(p), idx[p]))
            assert_same_leaves(got, canon_from_list(og.leaf_table(p)))
            assert_same_leaves(got, canon_from_list(views_table(d.get_leaf_points(p), idx[p])))
            assert [g.n_nodes(p), g.n_leaves(p), g.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]
            assert sorted(map(bytes, g.get_points(p))) == sorted(map(bytes, d.get_points(p)))

    for p in (0, 1):
        for x in (g, d, og):
            x.insert_points(p, poses[p])
    same([0, 1])
    with pytest.raises(ValueError, match="Cannot insert points to existing pose 1"):
        g.insert_points(1, poses[1])
    g.subdivide(crit(20), [0])
    d.subdivide(crit(20), [0])
    og.subdivide(20, [0])
    same([0, 1])
    for x in (g, d, og):
        x.insert_points(2, poses[2])      # a late pose inherits the scheme
    same([0, 1, 2])
    keep = [lambda pts: len(pts) >= 3]
    g.filter(keep)
    d.filter(keep)
    og.filter(keep)
    same([0, 1, 2])
    np.random.seed(4)
    table = np.random.random((128, 6))
    g.map_leaf_points_cuda_ransac(poses_per_batch=2, threshold=0.05, hypotheses=table)
    d.map_leaf_points_cuda_ransac(poses_per_batch=2, threshold=0.05, hypotheses=table)
    for p in range(3):
        assert sorted(map(bytes, g.get_points(p))) == sorted(map(bytes, d.get_points(p)))
        assert g.n_points(p) == d.n_points(p)
    # the plug types really were instantiated and called
    if plug in ("manager", "both"):
        assert calls["manager_init"] >= 8 and calls["manager_insert"] >= calls["manager_init"]
    if plug in ("octree", "both"):
        assert calls["octree_init"] >= 3 * 8 and calls["octree_subdivide_as"] >= 3 * 8


def test_manager_insert_points_into_any_existing_pose_vs_oracle():
    """OctreeManager.insert_points on a pose that already has an octree appends to it and lets the new
    points descend the current scheme (octree_manager.py:161-171) - for ANY pose, not only the most
    recently inserted one."""
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp

    rng = np.random.default_rng(91)
    poses = [rng.random((500, 3)), rng.random((400, 3)), rng.random((300, 3))]
    extra = {0: rng.random((151, 3)), 1: rng.random((80, 3))}
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(3):
        m.insert_points(p, poses[p])
        om.insert_points(p, poses[p])
    m.subdivide(crit(25))
    om.subdivide(25)
    for p in (0, 1):       # neither is the last pose; 151 points: an odd offset for the later poses
        m.insert_points(p, extra[p])
        om.insert_points(p, extra[p])
    allp = [np.vstack([poses[0], extra[0]]), np.vstack([poses[1], extra[1]]), poses[2]]
    for p in range(3):
        index = index_map(allp[p])
        got = canon_from_list(views_table(m.get_leaf_points(True, p), index))
        assert_same_leaves(got, canon_from_list(onp.tree_leaf_table(om.octrees[p])))
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == [om.n_nodes(p), om.n_leaves(p), om.n_points(p)]
    m.subdivide(crit(10))
    om.subdivide(10)
    for p in range(3):
        index = index_map(allp[p])
        got = canon_from_list(views_table(m.get_leaf_points(True, p), index))
        assert_same_leaves(got, canon_from_list(onp.tree_leaf_table(om.octrees[p])))


@pytest.mark.parametrize("n_extra", [300, 301, 10, 11])
def test_manager_extend_pose_with_more_points_than_the_later_poses_hold(n_extra):
    """The store is pose-major: points appended to an EARLIER pose are rotated in front of the later
    poses' points.  With more new points than later points (300 vs 10) the two pieces overlap in the
    store - the rotation must not copy store-to-store (round-2 advisor finding)."""
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp

    rng = np.random.default_rng(1000 + n_extra)
    poses = [rng.random((500, 3)), rng.random((10, 3))]
    extra = rng.random((n_extra, 3))
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(2):
        m.insert_points(p, poses[p])
        om.insert_points(p, poses[p])
    m.subdivide(crit(20))
    om.subdivide(20)
    m.insert_points(0, extra)
    om.insert_points(0, extra)
    allp = [np.vstack([poses[0], extra]), poses[1]]
    for rnd in range(2):
        for p in range(2):
            assert np.array_equal(np.sort(m.get_points(p), axis=0), np.sort(allp[p], axis=0))
            index = index_map(allp[p])
            got = canon_from_list(views_table(m.get_leaf_points(True, p), index))
            assert_same_leaves(got, canon_from_list(onp.tree_leaf_table(om.octrees[p])))
            assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == [om.n_nodes(p), om.n_leaves(p), om.n_points(p)]
        m.subdivide(crit(7))
        om.subdivide(7)


# ------------------------------------------------------------------------------------------------
# the two build paths (bucket build / level-synchronous) must produce identical tables
# ------------------------------------------------------------------------------------------------
def _canon_build(out):
    """A build result independent of the node numbering (the bucket build numbers level-major; where the
    level loop subdivides voxels it left behind, those nodes are appended): nodes keyed by (voxel, child
    digit path), blocks as (leaf key, slot, size) in storage order."""
    nd, blk = out[0], out[1]
    n = len(nd["parent"])
    keys = [None] * n
    order = np.argsort(nd["depth"], kind="stable")
    par, fc = nd["parent"], nd["first_child"]
    for i in order.tolist():
        p = par[i]
        keys[i] = (int(nd["voxel"][i]), ()) if p < 0 else (keys[p][0], keys[p][1] + (int(i - fc[p]),))
    nodes = {keys[i]: (int(nd["depth"][i]), nd["corner"][i].tobytes(), nd["edge"][i].tobytes(), int(nd["epoch"][i]),
                       bool(fc[i] >= 0)) for i in range(n)}
    assert len(nodes) == n
    blocks = [(keys[b], int(sl), int(st), int(sz)) for b, sl, st, sz in
              zip(blk["node"].tolist(), blk["slot"].tolist(), blk["start"].tolist(), blk["size"].tolist())]
    return nodes, blocks


def _assert_same_build(a, b):
    assert a[5] == b[5]
    na, ba = _canon_build(a)
    nb, bb = _canon_build(b)
    assert na == nb
    assert ba == bb
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("scheme", [None, [0]])
def test_voxel_local_build_equals_level_synchronous_build(monkeypatch, scheme):
    from octreelib_amd import synthetic
    from octreelib_amd._engine import Forest

    clouds = [synthetic.planar_cloud(120_000, (8, 8, 8), seed=3, stream=1),
              synthetic.planar_cloud(60_000, (8, 8, 8), seed=3, stream=2)]

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        for c in clouds:
            f.add_pose(c)
        f.subdivide(48, scheme)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out

    set_option("NO_BUCKET_BUILD", 0)
    a = build()
    set_option("NO_BUCKET_BUILD", 1)
    b = build()
    assert a[5] == b[5] and a[5] >= 2
    for k in ("voxel", "depth", "parent", "first_child", "corner", "edge", "epoch"):
        assert np.array_equal(a[0][k], b[0][k]), k
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k]), k
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("seed", range(10))
def test_voxel_local_build_random_voxels_vs_level_synchronous_build(monkeypatch, seed):
    """Random voxel populations around the limits of the voxel-local path (1 ... 512 points per
    voxel, K from 1, clusters that need more than its 7 levels and send the build down the general
    path): node tables, blocks, permutation, coordinates and leaf order must be bit-identical
    whichever path ran."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(500 + seed)
    dims = rng.integers(2, 7, 3)
    K = int(rng.choice([1, 2, 3, 8, 64, 511]))
    deep = seed % 3 == 2   # clusters tighter than 2^-7 of the voxel: the voxel-local path must fall back
    poses = []
    for _ in range(int(rng.integers(1, 4))):
        parts = []
        for c in np.argwhere(np.ones(dims)):
            r = rng.random()
            m = 0 if r < 0.1 else (170 if r < 0.2 else int(rng.integers(1, 171)))   # <= 512 over three poses
            pts = rng.random((m, 3))
            if m and rng.random() < 0.3:
                w = 2.0 ** -(12 if deep else 4)
                pts[: m // 2] = rng.random(3) * (1 - w) + rng.random((m // 2, 3)) * w
            parts.append(pts + c)
        cloud = np.unique(np.vstack(parts), axis=0)
        rng.shuffle(cloud)
        poses.append(cloud)
    scheme = None if rng.random() < 0.5 else sorted(rng.choice(len(poses), int(rng.integers(1, len(poses) + 1)), replace=False).tolist())
    ctx = nat.get_context()

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        for c in poses:
            f.add_pose(c)
        ctx.set_profiling(True)
        f.subdivide(K, scheme)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out, names

    set_option("NO_BUCKET_BUILD", 0)
    a, names_a = build()
    set_option("NO_BUCKET_BUILD", 1)
    b, names_b = build()
    assert "bucket_build" in names_a and "bucket_build" not in names_b
    if not deep and K >= 8:
        assert "level_hist" not in names_a    # the bucket path did the whole build
        for k in ("voxel", "depth", "parent", "first_child", "corner", "edge", "epoch"):
            assert np.array_equal(a[0][k], b[0][k]), k     # ... with the numbering of the general path
    _assert_same_build(a, b)


@pytest.mark.parametrize("per_voxel,K", [(490, 64), (470, 16), (498, 300), (250, 64)])
def test_bucket_build_nearly_full_buckets_vs_level_synchronous_build(monkeypatch, per_voxel, K):
    """Evenly filled scenes whose buckets of 8 (or 16) voxels hold 3 600 ... 4 200 points: the bucket kernel keeps
    the coordinates of its first seven rounds of 512 points in registers and reads the last round again at the
    output (buckets above 3 584 points), buckets just above 4 096 points go through the chunk plan.  Both must give
    the level-synchronous build bit for bit."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    # (the host sizes buckets for OCTL_BUCKET_POINTS points on average - 2 560 by default, so that an evenly filled
    #  scene stays well inside the capacity; 4 000 puts the buckets of this scene around it)
    set_option("BUCKET_POINTS", 4000)
    rng = np.random.default_rng(per_voxel + K)
    dims = (8, 8, 8)
    parts = []
    for c in np.argwhere(np.ones(dims)):
        m = int(rng.integers(per_voxel - 25, per_voxel + 26))
        parts.append(rng.random((m, 3)) + c)
    cloud = np.unique(np.vstack(parts), axis=0)
    rng.shuffle(cloud)
    ctx = nat.get_context()

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        f.add_pose(cloud)
        ctx.set_profiling(True)
        f.subdivide(K, None)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out, names

    set_option("NO_BUCKET_BUILD", 0)
    a, names_a = build()
    set_option("NO_BUCKET_BUILD", 1)
    b, names_b = build()
    assert "bucket_build" in names_a and "level_hist" not in names_a
    assert "bucket_build" not in names_b
    assert a[5] == b[5]
    for k in ("voxel", "depth", "parent", "first_child", "corner", "edge", "epoch"):
        assert np.array_equal(a[0][k], b[0][k]), k
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k]), k
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("seed", range(6))
def test_bucket_build_mixed_voxel_populations_vs_level_synchronous_build(monkeypatch, seed):
    """Skewed scenes: most voxels hold a few hundred points, some 600 .. 3000 (more than one wavefront's
    512, up to the 4096 a workgroup sorts in LDS), a few are empty; buckets of consecutive voxels
    overflow and are cut into chunks of whole voxels.  Node tables, blocks, permutation, coordinates and
    leaf order must be bit-identical to the level-synchronous path, and the bucket path must have done
    the whole build."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(900 + seed)
    dims = rng.integers(3, 7, 3)
    K = int(rng.choice([16, 64, 300, 1000]))
    parts = []
    for c in np.argwhere(np.ones(dims)):
        r = rng.random()
        m = 0 if r < 0.05 else (int(rng.integers(600, 3001)) if r < 0.30 else int(rng.integers(1, 400)))
        pts = rng.random((m, 3))
        if m and rng.random() < 0.3:
            pts[: m // 2] = rng.random(3) * 0.9 + rng.random((m // 2, 3)) * 0.1
        parts.append(pts + c)
    cloud = np.unique(np.vstack(parts), axis=0)
    rng.shuffle(cloud)
    cut = int(len(cloud) * rng.uniform(0.3, 0.7))
    poses = [cloud[:cut], cloud[cut:]] if seed % 2 else [cloud]
    scheme = [0] if (len(poses) == 2 and seed % 4 == 1) else None
    ctx = nat.get_context()

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        for c in poses:
            f.add_pose(c)
        ctx.set_profiling(True)
        f.subdivide(K, scheme)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out, names

    set_option("NO_BUCKET_BUILD", 0)
    a, names_a = build()
    set_option("NO_BUCKET_BUILD", 1)
    b, names_b = build()
    assert "bucket_build" in names_a and "level_hist" not in names_a
    assert "bucket_build" not in names_b
    assert a[5] == b[5]
    for k in ("voxel", "depth", "parent", "first_child", "corner", "edge", "epoch"):
        assert np.array_equal(a[0][k], b[0][k]), k
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k]), k
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("seed", range(3))
def test_bucket_build_two_pass_partition_vs_level_synchronous_build(monkeypatch, seed):
    """More than 4096 buckets (forced here on a small cloud over ~10 k voxels; at full size: more than
    ~10 M points per GPU, e.g. one rank of the 10^9-point configuration): the partition runs two stable
    passes and the bucket bounds come from the sorted records.  Everything must be bit-identical to the
    level-synchronous path."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(1200 + seed)
    dims = np.array([24, 20, 22]) + rng.integers(0, 4, 3)
    n = 150_000
    q = np.stack([rng.integers(0, dims[a], n) for a in range(3)], axis=1)
    keep = rng.random(n) < np.where((q.sum(axis=1) % 7) == 0, 1.0, 0.25)   # some voxels 4x denser
    cloud = np.unique((rng.random((n, 3)) + q)[keep] - np.array([3.0, 0.0, 5.0]), axis=0)
    rng.shuffle(cloud)
    poses = [cloud[: len(cloud) // 2], cloud[len(cloud) // 2:]]
    K = int(rng.choice([3, 8, 20]))
    ctx = nat.get_context()

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        for c in poses:
            f.add_pose(c)
        ctx.set_profiling(True)
        f.subdivide(K, [1] if seed == 1 else None)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels), f.voxels.copy())
        f.close()
        return out, names

    set_option("NO_BUCKET_BUILD", 0)
    set_option("BUCKET_POINTS", 4)
    a, names_a = build()
    set_option("BUCKET_POINTS", 0)
    set_option("NO_BUCKET_BUILD", 1)
    b, names_b = build()
    assert "bucket_bounds" in names_a and "level_hist" not in names_a   # two passes, whole build
    assert "bucket_build" not in names_b
    assert a[5] == b[5] and np.array_equal(a[6], b[6])
    for k in ("voxel", "depth", "parent", "first_child", "corner", "edge", "epoch"):
        assert np.array_equal(a[0][k], b[0][k]), k
    for k in a[1]:
        assert np.array_equal(a[1][k], b[1][k]), k
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])


@pytest.mark.parametrize("seed", range(4))
def test_bucket_build_leaves_huge_and_deep_voxels_to_the_level_loop(monkeypatch, seed):
    """A few voxels hold more points than a workgroup sorts in LDS (5000 .. 9000) or clusters that need
    more than 6 levels: the bucket build finishes every other voxel and the level loop subdivides exactly
    those (it must NOT redo the whole build).  Same tables as the level-synchronous path."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(1500 + seed)
    dims = rng.integers(4, 8, 3)
    K = int(rng.choice([24, 64, 200]))
    parts, special = [], 0
    for c in np.argwhere(np.ones(dims)):
        r = rng.random()
        if r < 0.03:
            m = int(rng.integers(5000, 9001))         # beyond the 4096 points of one workgroup
            pts = rng.random((m, 3))
            special += 1
        elif r < 0.06:
            m = int(rng.integers(200, 600))
            pts = rng.random((m, 3))
            pts[: m // 2] = rng.random(3) * 0.9 + rng.random((m // 2, 3)) * 2.0 ** -9   # needs > 6 levels
            special += 1
        else:
            m = int(rng.integers(0, 500))
            pts = rng.random((m, 3))
        parts.append(pts + c)
    assert special >= 2
    cloud = np.unique(np.vstack(parts), axis=0)
    rng.shuffle(cloud)
    cut = int(len(cloud) * 0.6)
    poses = [cloud[:cut], cloud[cut:]]
    scheme = [1] if seed == 2 else None
    ctx = nat.get_context()

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        for c in poses:
            f.add_pose(c)
        ctx.set_profiling(True)
        f.subdivide(K, scheme)
        t = ctx.timings()
        ctx.set_profiling(False)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out, t

    set_option("NO_BUCKET_BUILD", 0)
    a, ta = build()
    set_option("NO_BUCKET_BUILD", 1)
    b, tb = build()
    assert "bucket_build" in ta and "level_hist" in ta and "keygen" not in ta    # resumed, not redone
    assert "bucket_build" not in tb and "keygen" in tb
    _assert_same_build(a, b)


@pytest.mark.parametrize("host_copy", [False, True])
def test_grid_get_points_row_order_golden(host_copy):
    """Grid.get_points against the reference's own row order (tests/golden/grid_get_points.npz, generated by the
    reference: two poses whose voxels are created in different orders, before and after a subdivide): the SEQUENCE of
    leaves along the rows must be the reference's (managers in creation order, grid.py:234-242, leaves depth first,
    octree.py:55-65) and every leaf must contribute its rows - inside a leaf the reference's order is an artefact of
    an unstable argsort.  host_copy: with the whole ordered cloud already on the host the rows are sliced there,
    otherwise they come from one device gather (octl_forest_gather_blocks)."""
    from octreelib_amd.grid import Grid, GridConfig

    gold = load_golden("grid_get_points.npz")
    poses = [gold["points0"], gold["points1"]]
    grid = Grid(GridConfig(voxel_edge_length=1))
    for p, pts in enumerate(poses):
        grid.insert_points(p, pts)

    def runs(a):
        a = np.asarray(a)
        return a[np.concatenate(([True], a[1:] != a[:-1]))]

    for stage, K in (("pre", None), ("k40", 40)):
        if K is not None:
            grid.subdivide(crit(K))
        for p, pts in enumerate(poses):
            index = {pts[i].tobytes(): i for i in range(len(pts))}
            f = grid._forest
            if host_copy:
                f.xyz          # (fetches and keeps the ordered cloud)
            else:
                f._xyz = None
            rows = np.ascontiguousarray(grid.get_points(p))
            assert (f._xyz is not None) == host_copy
            idx = np.array([index[rows[i].tobytes()] for i in range(len(rows))], dtype=np.int64)
            # every row once
            assert np.array_equal(np.sort(idx), np.sort(gold[f"{stage}_p{p}_rows_idx"]))
            # leaf of every original index, from the reference's leaf table (leaf = position in get_leaf_points' list)
            sizes = gold[f"{stage}_p{p}_sizes"]
            leaf_of = np.empty(len(pts), dtype=np.int64)
            leaf_of[gold[f"{stage}_p{p}_idx"]] = np.repeat(np.arange(len(sizes)), sizes)
            want_seq = runs(gold[f"{stage}_p{p}_rows_leaf"])
            got_seq = runs(leaf_of[idx])
            assert np.array_equal(got_seq, want_seq)
            assert len(np.unique(got_seq)) == len(got_seq)          # a leaf's rows are contiguous


def test_grid_get_points_follows_voxel_creation_order():
    """Grid.get_points walks ALL managers in the order they were first created (the dict order of
    Grid.__octrees, grid.py:56,100-109,240-242), not in lexicographic voxel order: a voxel first
    touched by a later pose comes after every voxel of the earlier poses."""
    from octreelib_amd.grid import Grid, GridConfig

    rng = np.random.default_rng(8)
    in_voxel = lambda q, m: rng.random((m, 3)) * 0.9 + np.asarray(q, dtype=float)  # noqa: E731
    a, b, c, d = (0, 0, 0), (0, 1, 0), (2, 0, 0), (1, 5, 1)      # a < b < d < c lexicographically
    p0 = np.vstack([in_voxel(c, 5), in_voxel(b, 4)])              # creates b, c (np.unique order)
    p1 = np.vstack([in_voxel(c, 3), in_voxel(a, 6), in_voxel(d, 2)])   # creates a, d afterwards
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, p0)
    grid.insert_points(1, p1)
    # creation order: b, c, a, d
    assert np.array_equal(grid.get_points(0), np.vstack([p0[5:], p0[:5]]))
    assert np.array_equal(grid.get_points(1), np.vstack([p1[:3], p1[3:9], p1[9:]]))
    grid.subdivide(crit(2))
    got = grid.get_points(1)
    vox = np.floor(got).astype(int)
    firsts = [tuple(v) for i, v in enumerate(vox.tolist()) if i == 0 or vox[i].tolist() != vox[i - 1].tolist()]
    assert firsts == [c, a, d]                                    # still creation order after subdivide
    assert sorted(map(bytes, got)) == sorted(map(bytes, p1))
    # leaves, in contrast, are listed in lexicographic voxel order (grid.py:217-232)
    lv = [tuple(np.floor(np.asarray(v.corner_min, dtype=float)).astype(int).tolist()) for v in grid.get_leaf_points(1)]
    assert [k for i, k in enumerate(lv) if i == 0 or lv[i - 1] != k] == [a, d, c]


def test_octree_subdivide_as_between_stand_alone_octrees_vs_oracle():
    """Octree.subdivide_as(other) (octree.py:222-227): structure, leaf order (history dependent) and
    counters against the oracle, incl. a second call with a finer scheme."""
    from octreelib_amd.octree import Octree, OctreeConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(12)
    corner, edge = np.array([0.0, 0.0, 0.0]), np.float64(2)
    pa = rng.random((3000, 3)) * 2.0
    pb = np.vstack([rng.random((1500, 3)) * 2.0, 0.3 + rng.random((2500, 3)) * 0.2])
    a, b = Octree(OctreeConfig(), corner, edge), Octree(OctreeConfig(), corner, edge)
    oa, ob = onp.OTree(corner, edge), onp.OTree(corner, edge)
    for t, o, p in ((a, oa, pa), (b, ob, pb)):
        t.insert_points(p)
        o.insert_points(p)
    b.subdivide(crit(100))
    ob.subdivide(100)
    a.subdivide(crit(900))          # a has its own coarser history first
    oa.subdivide(900)
    a.subdivide_as(b)
    oa.subdivide_as(ob)
    ia = index_map(pa)
    assert_same_leaves(canon_from_list(views_table(a.get_leaf_points(), ia)), canon_from_list(onp.tree_leaf_table(oa)))
    assert [a.n_nodes, a.n_leaves, a.n_points] == [oa.n_nodes, oa.n_leaves, oa.n_points]
    assert a.n_nodes == b.n_nodes
    # a finer scheme from another octree (subdividing b again would change nothing: its root holds
    # no points any more, octree.py:20-32)
    b2, ob2 = Octree(OctreeConfig(), corner, edge), onp.OTree(corner, edge)
    b2.insert_points(pb)
    ob2.insert_points(pb)
    b2.subdivide(crit(30))
    ob2.subdivide(30)
    assert b2.n_nodes > b.n_nodes
    a.subdivide_as(b2)
    oa.subdivide_as(ob2)
    b, ob = b2, ob2
    assert_same_leaves(canon_from_list(views_table(a.get_leaf_points(), ia)), canon_from_list(onp.tree_leaf_table(oa)))
    assert [a.n_nodes, a.n_leaves, a.n_points] == [oa.n_nodes, oa.n_leaves, oa.n_points]
    # an empty octree takes the structure too
    c, oc = Octree(OctreeConfig(), corner, edge), onp.OTree(corner, edge)
    c.subdivide_as(b)
    oc.subdivide_as(ob)
    assert [c.n_nodes, c.n_leaves, c.n_points] == [oc.n_nodes, oc.n_leaves, oc.n_points]
    with pytest.raises(ValueError):
        Octree(OctreeConfig(), corner + 1.0, edge).subdivide_as(b)


def test_forest_is_usable_after_a_failed_subdivide():
    from octreelib_amd.grid import Grid, GridConfig

    rng = np.random.default_rng(3)
    pts = np.vstack([rng.random((500, 3)), np.tile(np.array([[0.25, 0.25, 0.25]]), (4, 1))])
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    grid.subdivide(crit(100))
    assert grid.n_points(0) == 504
    with pytest.raises(RecursionError):
        grid.subdivide(crit(2))  # four identical points never separate
    # the scheme is gone, the points are not
    assert [grid.n_leaves(0), grid.n_points(0), grid.n_nodes(0)] == [1, 504, 1]
    grid.subdivide(crit(100))
    assert grid.n_points(0) == 504 and grid.n_leaves(0) > 1


def test_integration_md_binding_stub_runs():
    """The reference-side ctypes binding shown in INTEGRATION.md (Option B) is executed verbatim
    (only the library path is made absolute) and must return the oracle's mask."""
    import os
    import re

    from octreelib_amd import _native as nat
    from oracle import ransac_np as rnp

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# octreelib/ransac/hip_ransac.py.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("liboctree_hip.so")', f'C.CDLL({nat.lib_path()!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md:hip_ransac.py", "exec"), ns)
    rng = np.random.default_rng(2)
    sizes = rng.integers(0, 50, 80).astype(np.int32)
    cloud = rng.random((int(sizes.sum()), 3))
    cloud[:, 2] = 0.3 * cloud[:, 0] + rng.normal(0, 0.004, len(cloud))
    np.random.seed(42)
    op = ns["HipRansac"](threshold=0.01, hypotheses_number=512, initial_points_number=6)
    mask = op.evaluate(cloud, sizes)
    assert mask.dtype == np.bool_ and mask.shape == (len(cloud),)
    assert np.array_equal(mask, rnp.evaluate(cloud, sizes, op._table, 0.01))


def _incremental_poses():
    rng = np.random.default_rng(23)
    poses = [rng.random((20_000, 3)) * 6.0,                                  # the scheme's pose
             rng.random((6_000, 3)) * 6.0,                                   # known voxels only
             rng.random((7_000, 3)) * 6.0 + np.array([4.0, 0.0, -3.0]),      # new voxels in front of and behind the old
             rng.random((3_000, 3)) * 2.0 + np.array([-5.0, 9.0, 1.0]),      # new voxels only
             rng.random((5_000, 3)) * 8.0 - 1.0]
    return poses


def test_incremental_insertion_vs_oracle():
    """Poses appended to a subdivided grid are placed into the existing scheme point by point
    (incremental.hip) - known voxels, new voxels on both sides of the old ones, two poses between two
    queries - then RANSAC-free refinement: leaf tables and counters against the oracle at every step."""
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    poses = _incremental_poses()
    idx = [index_map(p) for p in poses]
    grid, og = Grid(GridConfig(voxel_edge_length=2)), onp.OGrid(2)

    def check(n):
        for p in range(n):
            assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), _oracle_pose_table(og, p))
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]

    grid.insert_points(0, poses[0])
    og.insert_points(0, poses[0])
    grid.subdivide(crit(100))
    og.subdivide(100)
    check(1)
    for p in (1, 2):
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])
        check(p + 1)
    for p in (3, 4):   # two poses before the next query
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])
    check(5)
    grid.subdivide(crit(40))
    og.subdivide(40)
    check(5)


def test_incremental_insertion_equals_replacement(monkeypatch):
    """The same sequence through the incremental path and through the re-placement of every stored point
    (keep_scheme build of build.hip): same scheme, and per (leaf, pose) block the same points in the same
    order, listed in the same reference order; only the storage order of the blocks differs."""
    from octreelib_amd._engine import Forest

    poses = _incremental_poses()

    def run():
        f = Forest(0, np.zeros(3), 2.0)
        f.add_pose(poses[0])
        f.subdivide(100)
        f.ctx.sync()
        f.ctx.set_profiling(True)
        snaps = []
        for group in ((1,), (2,), (3, 4)):
            for p in group:
                f.add_pose(poses[p])
            f.ensure_built()
            f.ctx.sync()
            nd = {k: v.copy() for k, v in f.nodes.items()}
            blk = {k: v.copy() for k, v in f.blocks.items()}
            nodes, blocks = _canon_build((nd, blk))
            xyz, perm, order = f.xyz, f.perm, f.order
            listing = [(blocks[b][0], blocks[b][1], xyz[blocks[b][2]:blocks[b][2] + blocks[b][3]].tobytes(),
                        perm[blocks[b][2]:blocks[b][2] + blocks[b][3]].tobytes()) for b in order.tolist()]
            snaps.append((nodes, listing, f.voxels.copy(), [f.n_leaves(s) for s in range(f.n_slots)],
                          [f.n_nodes(s) for s in range(f.n_slots)]))
        timers = f.ctx.timings()
        f.ctx.set_profiling(False)
        # RANSAC over everything, then removal of the outliers: the compaction keeps the storage order
        np.random.seed(4)
        table = np.random.random((256, 6))
        f.ransac_all(10, table, 0.05)
        f.apply_device_mask()
        blk = {k: v.copy() for k, v in f.blocks.items()}
        nodes, blocks = _canon_build(({k: v.copy() for k, v in f.nodes.items()}, blk))
        xyz, order = f.xyz, f.order
        after = [(blocks[b][0], blocks[b][1], xyz[blocks[b][2]:blocks[b][2] + blocks[b][3]].tobytes())
                 for b in order.tolist()]
        f.close()
        return snaps, after, timers

    set_option("NO_INCREMENTAL", 0)
    a, a_after, ta = run()
    set_option("NO_INCREMENTAL", 1)
    b, b_after, tb = run()
    # the first run placed only the new points (three insertions, two of them with new voxels), the
    # second re-placed everything
    assert ta["inc_place"][1] == 3 and ta["inc_new_voxels"][1] == 2 and "keygen" not in ta
    assert "inc_place" not in tb and tb["keygen"][1] == 3
    for (na, la, va, leaves_a, nodes_a), (nb, lb, vb, leaves_b, nodes_b) in zip(a, b):
        assert na == nb
        assert la == lb
        assert np.array_equal(va, vb)
        assert leaves_a == leaves_b and nodes_a == nodes_b
    assert a_after == b_after


@pytest.mark.parametrize("empty_voxel", [False, True])
def test_bucket_build_over_a_previous_scheme_equals_level_synchronous_build(monkeypatch, empty_voxel):
    """subdivide on an already subdivided forest: the bucket path takes it too - nodes that were internal
    before keep their epochs (they decide the cached-leaf order), every voxel of the previous scheme has to
    be there again.  Same scheme (epochs included), same blocks in the same reference order as the
    level-synchronous path; a voxel that has lost all its points sends the build down that path."""
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(77)
    poses = [rng.random((30_000, 3)) * 5.0 - 1.0, rng.random((20_000, 3)) * 5.0 - 1.0, rng.random((25_000, 3)) * 6.0 - 1.5]

    def run():
        f = Forest(0, np.zeros(3), 1.0)
        f.add_pose(poses[0])
        f.add_pose(poses[1])
        f.subdivide(120, [0])
        f.add_pose(poses[2])                 # inherits the scheme (incremental placement)
        np.random.seed(3)
        f.ransac_all(10, np.random.random((256, 6)), 0.05)
        f.apply_device_mask()                # points leave the tree
        if empty_voxel:                      # ... and one voxel loses all of them
            v0 = int(f.nodes["voxel"][f.blocks["node"][0]])
            keep = np.ones(f.n_ord, dtype=np.uint8)
            for b in np.nonzero(f.nodes["voxel"][f.blocks["node"]] == v0)[0]:
                s0, z = int(f.blocks["start"][b]), int(f.blocks["size"][b])
                keep[s0:s0 + z] = 0
            f.apply_host_mask(keep)
        f.ctx.sync()
        f.ctx.set_profiling(True)
        f.subdivide(40)                      # all poses, finer: a build over the previous scheme
        f.ctx.sync()
        names = set(f.ctx.timings())
        f.ctx.set_profiling(False)
        nodes, blocks = _canon_build(({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()}))
        xyz, perm = f.xyz, f.perm
        listing = [(blocks[b][0], blocks[b][1], xyz[blocks[b][2]:blocks[b][2] + blocks[b][3]].tobytes(),
                    perm[blocks[b][2]:blocks[b][2] + blocks[b][3]].tobytes()) for b in f.order.tolist()]
        counts = [(f.n_nodes(s), f.n_leaves(s), f.n_points(s)) for s in range(f.n_slots)]
        vox = f.voxels.copy()
        f.close()
        return nodes, listing, counts, vox, names

    set_option("NO_BUCKET_HISTORY", 0)
    a = run()
    set_option("NO_BUCKET_HISTORY", 1)
    b = run()
    assert "bucket_build" in a[4] and "bucket_build" not in b[4] and "keygen" in b[4]
    assert ("keygen" in a[4]) == empty_voxel    # the empty voxel: back to the level-synchronous path
    assert a[0] == b[0]
    assert a[1] == b[1]
    assert a[2] == b[2]
    assert np.array_equal(a[3], b[3])
    assert len({e for (_d, _c, _e, e, _i) in a[0].values()}) > 1    # epochs of two builds in the scheme


@pytest.mark.parametrize("L", [2.0, 3.0, 5.0, 7.0])
def test_bucket_build_integer_edges_vs_level_synchronous_build_and_oracle(monkeypatch, L):
    """Voxel edges that are not 1 (the general floor-division of grid.py:72-76 instead of floor(), child
    edges L / 2^j that are not powers of two), negative coordinates, two poses: the bucket path against
    the level-synchronous path bit for bit, and a reduced case against the oracle."""
    from octreelib_amd._engine import Forest
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(int(L) * 11)
    poses = [(rng.random((120_000, 3)) - 0.45) * 9.0 * L, (rng.random((80_000, 3)) - 0.45) * 9.0 * L]
    # points exactly on voxel and child faces, and one ulp to either side of them
    faces = np.arange(-4, 5)[:, None] * np.array([L, L / 2, L / 4])[None, :]
    edge_pts = np.stack(np.meshgrid(faces[:, 0], faces[:, 1], faces[:, 2]), -1).reshape(-1, 3)
    below = np.nextafter(edge_pts, -np.inf)
    # (just below zero the reference itself fails: p - corner rounds to the full edge; the library
    #  raises its DomainError there, tested elsewhere)
    below = below[~((below < 0) & (below > -1e-300)).any(axis=1)]
    poses[0] = np.unique(np.vstack([poses[0], edge_pts, np.nextafter(edge_pts, np.inf), below]), axis=0)
    rng.shuffle(poses[0])

    def build():
        f = Forest(0, np.zeros(3), L)
        for c in poses:
            f.add_pose(c)
        f.subdivide(30, [0])
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out

    set_option("NO_BUCKET_BUILD", 0)
    a = build()
    set_option("NO_BUCKET_BUILD", 1)
    b = build()
    set_option("NO_BUCKET_BUILD", 0)
    _assert_same_build(a, b)
    # reduced: against the oracle through the drop-in classes
    small = [p[:6000] for p in poses]
    grid, og = Grid(GridConfig(voxel_edge_length=int(L))), onp.OGrid(int(L))
    for p, c in enumerate(small):
        grid.insert_points(p, c)
        og.insert_points(p, c)
    grid.subdivide(crit(20), [0])
    og.subdivide(20, [0])
    for p, c in enumerate(small):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index_map(c))), _oracle_pose_table(og, p))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]


def test_bucket_build_far_from_the_origin_vs_level_synchronous_build_and_oracle(monkeypatch):
    """Voxel indices near the limit of the domain (+-2^20): the differences p - corner are formed at
    magnitudes where an ulp is 1e-10 of a voxel - both build paths and the oracle must still agree."""
    from octreelib_amd._engine import Forest
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(5150)
    off = np.array([1_000_000.0, -1_040_000.0, 524_287.0])
    cloud = rng.random((150_000, 3)) * np.array([6.0, 5.0, 4.0]) + off
    cloud[:40_000] = off + rng.random(3) * 3 + rng.random((40_000, 3)) * 0.02    # a dense cluster: deep trees
    cloud = np.unique(cloud, axis=0)
    rng.shuffle(cloud)

    def build():
        f = Forest(0, np.zeros(3), 1.0)
        f.add_pose(cloud)
        f.subdivide(24)
        out = ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()},
               f.perm.copy(), f.xyz.copy(), f.order.copy(), int(f.info.n_levels))
        f.close()
        return out

    set_option("NO_BUCKET_BUILD", 0)
    a = build()
    set_option("NO_BUCKET_BUILD", 1)
    b = build()
    set_option("NO_BUCKET_BUILD", 0)
    _assert_same_build(a, b)
    assert a[5] >= 5
    small = cloud[:8000]
    grid, og = Grid(GridConfig(voxel_edge_length=1)), onp.OGrid(1)
    grid.insert_points(0, small)
    og.insert_points(0, small)
    grid.subdivide(crit(10))
    og.subdivide(10)
    assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(0), index_map(small))), _oracle_pose_table(og, 0))
    assert [grid.n_nodes(0), grid.n_leaves(0), grid.n_points(0)] == [og.n_nodes(0), og.n_leaves(0), og.n_points(0)]


@pytest.mark.parametrize("poses_per_batch", [1, 2])
def test_grid_ransac_batches_golden(poses_per_batch):
    """map_leaf_points_cuda_ransac over batches of poses on a subdivided grid, against the reference's own
    result (tests/golden/grid_ransac_batches.npz: three poses, K = 30, H = 128)."""
    from octreelib_amd.grid import Grid, GridConfig

    g = load_golden("grid_ransac_batches.npz")
    grid = Grid(GridConfig(voxel_edge_length=int(g["L"])))
    idx = []
    for p in range(3):
        grid.insert_points(p, g[f"points{p}"])
        idx.append(index_map(g[f"points{p}"]))
    grid.subdivide(crit(int(g["K"])))
    np.random.seed(int(g["seed"]))
    grid.map_leaf_points_cuda_ransac(poses_per_batch=poses_per_batch, threshold=0.01, hypotheses_number=128,
                                     initial_points_number=6)
    tag = f"b{poses_per_batch}"
    for p in range(3):
        assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"{tag}_p{p}"))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"{tag}_p{p}_counts"])


def test_grid_filter_golden():
    """Grid.filter with point-count criteria (the device kernel over the block table) against the
    reference's own leaf tables, two filters in a row."""
    from octreelib_amd import _native as nat
    from octreelib_amd.grid import Grid, GridConfig

    g = load_golden("grid_filter.npz")
    grid = Grid(GridConfig(voxel_edge_length=1))
    idx = []
    for p in range(2):
        grid.insert_points(p, g[f"points{p}"])
        idx.append(index_map(g[f"points{p}"]))
    grid.subdivide(crit(int(g["K"])))
    ctx = nat.get_context()
    for tag, criteria in (("ge5", [lambda pts: len(pts) >= 5]),
                          ("in3to12", [lambda pts: len(pts) > 2, lambda pts: 12 >= len(pts)])):
        ctx.set_profiling(True)
        grid.filter(criteria)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        assert "filter" in names    # the device path
        for p in range(2):
            assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"{tag}_p{p}"))
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"{tag}_p{p}_counts"])


def test_manager_extend_golden():
    """OctreeManager.insert_points into poses that already exist (neither the last one), then a finer
    subdivide: the reference's own leaf tables."""
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager

    g = load_golden("manager_extend.npz")
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(3):
        m.insert_points(p, g[f"points{p}"])
    m.subdivide(crit(25))
    for p in (0, 1):
        m.insert_points(p, g[f"extra{p}"])
    allp = [np.vstack([g["points0"], g["extra0"]]), np.vstack([g["points1"], g["extra1"]]), g["points2"]]
    idx = [index_map(a) for a in allp]

    def check(tag):
        for p in range(3):
            got = canon_from_list(views_table(m.get_leaf_points(True, p), idx[p]))
            assert_same_leaves(got, golden_canon(g, f"{tag}_p{p}"))
            assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    check("a")
    m.subdivide(crit(10))
    check("b")


def test_octree_subdivide_as_golden():
    """Octree.subdivide_as between stand-alone octrees against the reference's own leaf tables (structure,
    history-dependent leaf order, counters)."""
    from octreelib_amd.octree import Octree, OctreeConfig

    g = load_golden("octree_subdivide_as.npz")
    corner, edge = np.array([0.0, 0.0, 0.0]), np.float64(g["edge"])
    a, b, b2 = (Octree(OctreeConfig(), corner, edge) for _ in range(3))
    a.insert_points(g["pa"])
    b.insert_points(g["pb"])
    b2.insert_points(g["pb"])
    b.subdivide(crit(100))
    b2.subdivide(crit(30))
    a.subdivide(crit(900))
    ia = index_map(g["pa"])
    for tag, other in (("as100", b), ("as30", b2)):
        a.subdivide_as(other)
        assert_same_leaves(canon_from_list(views_table(a.get_leaf_points(), ia)), golden_canon(g, tag))
        assert [a.n_nodes, a.n_leaves, a.n_points] == list(g[f"{tag}_counts"])


def test_grid_callable_criterion_golden():
    """An arbitrary host callable as subdivision criterion (evaluated by the host level by level, placement
    on the device), then a count criterion on top: the reference's own leaf tables."""
    from octreelib_amd.grid import Grid, GridConfig

    def spread(points):
        return len(points) > 40 and float(points.max(axis=0).max() - points.min(axis=0).min()) > 0.3

    g = load_golden("grid_callable.npz")
    grid = Grid(GridConfig(voxel_edge_length=1))
    idx = []
    for p in range(2):
        grid.insert_points(p, g[f"points{p}"])
        idx.append(index_map(g[f"points{p}"]))

    def check(tag):
        for p in range(2):
            assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), idx[p])), golden_canon(g, f"{tag}_p{p}"))
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    grid.subdivide([spread], [0])
    check("spread")
    grid.subdivide(crit(15))
    check("k15")


def test_two_host_synchronisations_per_step():
    """clear + add_pose_device + build + ransac_all + apply_mask on a device-resident cloud: the host waits
    for the device twice (bucket totals; kept points and blocks), counted by the library itself."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic

    ctx = nat.get_context()
    lib = ctx.lib
    pts = np.ascontiguousarray(synthetic.planar_cloud(400_000, (12, 12, 12), seed=2))
    d = C.c_void_p()
    ctx.check(lib.octl_dev_alloc(ctx.handle, pts.nbytes, C.byref(d)))
    ctx.check(lib.octl_dev_upload(ctx.handle, d, nat.ptr(pts), pts.nbytes))
    fh = C.c_void_p()
    corner = np.zeros(3)
    ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fh)))
    np.random.seed(0)
    table = np.ascontiguousarray(np.random.random((1024, 6)))
    info, slot, n_alive = nat.BuildInfo(), C.c_int32(0), C.c_int64(0)

    def step():
        ctx.check(lib.octl_forest_clear(fh))
        ctx.check(lib.octl_forest_add_pose_device(fh, d, len(pts), C.byref(slot)))
        ctx.check(lib.octl_forest_build(fh, 64, None, 0, 0, 0, C.byref(info)))
        ctx.check(lib.octl_forest_ransac_all(fh, 10, None, 0, nat.ptr(table), 1024, 6, 0.01))
        ctx.check(lib.octl_forest_apply_mask(fh, C.byref(n_alive)))

    try:
        step()
        step()   # (the first steps allocate: growing a buffer waits for the stream)
        c0, c1 = C.c_uint64(0), C.c_uint64(0)
        ctx.check(lib.octl_debug_host_syncs(C.byref(c0)))
        for _ in range(3):
            step()
        ctx.check(lib.octl_debug_host_syncs(C.byref(c1)))
        assert c1.value - c0.value == 6
        assert 0 < n_alive.value < len(pts)
    finally:
        lib.octl_forest_destroy(fh)
        lib.octl_dev_free(ctx.handle, d)


@pytest.mark.parametrize("seed", range(8))
def test_block_order_left_by_the_bucket_build_equals_order_hip(monkeypatch, seed):
    """One pose, a fresh scheme: k_bucket_finish leaves the blocks' listing order behind (preorder ranks and
    block keys ranked inside each bucket); it must be the order order.hip computes from the node table -
    shallow and deep trees, skewed voxels, K from 1."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(4200 + seed)
    dims = rng.integers(2, 9, 3)
    K = int(rng.choice([1, 3, 8, 40, 64, 200]))
    parts = []
    for c in np.argwhere(np.ones(dims)):
        m = int(rng.choice([0, 3, 40, 170, 600]))
        pts = rng.random((m, 3))
        if m and rng.random() < 0.4:
            w = 2.0 ** -int(rng.integers(2, 6))
            pts[: m // 2] = rng.random(3) * (1 - w) + rng.random((m // 2, 3)) * w
        parts.append(pts + c - 2.0)
    cloud = np.unique(np.vstack(parts), axis=0)
    rng.shuffle(cloud)
    ctx = nat.get_context()

    def run():
        f = Forest(0, np.zeros(3), 1.0)
        f.add_pose(cloud)
        f.subdivide(K)
        ctx.sync()
        ctx.set_profiling(True)
        order = f.order.copy()
        names = set(ctx.timings())
        ctx.set_profiling(False)
        blocks = {k: v.copy() for k, v in f.blocks.items()}
        depth = int(f.info.max_depth)
        f.close()
        return order, blocks, names, depth

    set_option("NO_FAST_ORDER", 0)
    a, ba, na, depth = run()
    set_option("NO_FAST_ORDER", 1)
    b, bb, nb_, _ = run()
    for k in ba:
        assert np.array_equal(ba[k], bb[k]), k
    assert np.array_equal(a, b)
    assert sorted(a.tolist()) == list(range(len(a)))
    assert "ransac_order" in nb_                 # order.hip ran for the second forest ...
    if K >= 40 and depth <= 6:
        # ... and not for the first (a small K may exceed what a bucket ranks in LDS, a tree deeper than the
        # bucket kernel's six levels is finished by the level loop: both leave the order to order.hip)
        assert "ransac_order" not in na


# ------------------------------------------------------------------------------------------------
# a cloud read in place (octl_forest_add_pose_adopt) and the hinted key geometry
# ------------------------------------------------------------------------------------------------
def _device_cloud(ctx, pts):
    import ctypes as C

    from octreelib_amd import _native as nat

    d = C.c_void_p()
    ctx.check(ctx.lib.octl_dev_alloc(ctx.handle, max(pts.nbytes, 16), C.byref(d)))
    ctx.check(ctx.lib.octl_dev_upload(ctx.handle, d, nat.ptr(pts), pts.nbytes))
    return d


def _tables(f):
    return ({k: v.copy() for k, v in f.nodes.items()}, {k: v.copy() for k, v in f.blocks.items()}, f.perm.copy(),
            f.voxels.copy(), f.order.copy())


def _assert_same_tables(a, b):
    for x, y in zip(a[:2], b[:2]):
        assert x.keys() == y.keys()
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    for x, y in zip(a[2:], b[2:]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("two_pass", [False, True])
def test_adopted_cloud_builds_the_same_forest_with_and_without_a_geometry_hint(monkeypatch, two_pass):
    """octl_forest_add_pose_adopt reads the caller's device buffer in place; the voxel box of such a cloud is
    found by the build's own histogram pass under the geometry of the context's previous build (hit: same
    scene; miss: a cloud that leaves the hinted box, a box of another size) or by the box pass (no hint).
    Every variant must build exactly the tables of the copying path.  two_pass: many small buckets over more
    than 4096 voxel keys - the two-pass partition of the large clouds (one rank's 125 M points of BASELINE config
    5), where the HOST forms the geometry from the hint's box instead of waiting for the box pass."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic
    from octreelib_amd._engine import Forest

    ctx = nat.get_context()
    K = 48
    dm = (20, 20, 20) if two_pass else (8, 8, 8)
    if two_pass:
        set_option("BUCKET_POINTS", 16)
    scenes = {
        "base": synthetic.planar_cloud(200_000, dm, seed=1, stream=0),
        "same_box": synthetic.planar_cloud(200_000, dm, seed=1, stream=1),
        "shifted": synthetic.planar_cloud(200_000, dm, seed=1, stream=2) + np.array([3.0, -2.0, 0.0]),
        "grown": np.vstack([synthetic.planar_cloud(199_990, dm, seed=1, stream=3), np.full((10, 3), 40.5) + np.arange(10)[:, None] * 1e-3]),
        "shrunk": synthetic.planar_cloud(200_000, dm, seed=1, stream=4, box=((2, 2, 2), (5, 5, 5))),
        "negative": synthetic.planar_cloud(200_000, dm, seed=1, stream=5) - 100.0,
    }

    def reference(pts):
        f = Forest(0, np.zeros(3), 1.0)
        f.add_pose(pts)
        f.subdivide(K)
        t = _tables(f)
        f.close()
        return t

    want = {k: reference(v) for k, v in scenes.items()}
    used_box_pass = []
    for name in ["base", "same_box", "same_box", "shifted", "base", "grown", "shrunk", "negative", "base"]:
        pts = scenes[name]
        d = _device_cloud(ctx, pts)
        g = Forest(0, np.zeros(3), 1.0)
        ctx.set_profiling(True)
        g.add_pose_device(d, len(pts), adopt=True)
        g.subdivide(K)
        ctx.sync()
        tm = ctx.timings()
        used_box_pass.append("ingest" in tm)
        if name in ("base", "same_box"):
            assert ("bucket_bounds" in tm) == two_pass      # (the partition really took the path under test)
        ctx.set_profiling(False)
        _assert_same_tables(_tables(g), want[name])
        # RANSAC + apply_mask + a second subdivide on the adopted store (alive flags in play)
        np.random.seed(0)
        table = np.random.random((256, 6))
        g.ransac_all(10, table, 0.01)
        g.apply_device_mask()
        g.subdivide(K // 2)
        h = Forest(0, np.zeros(3), 1.0)
        h.add_pose(pts)
        h.subdivide(K)
        h.ransac_all(10, table, 0.01)
        h.apply_device_mask()
        h.subdivide(K // 2)
        _assert_same_tables(_tables(g), _tables(h))
        g.close()
        h.close()
        ctx.check(ctx.lib.octl_dev_free(ctx.handle, d))
    # the first build of the context's life may need the box pass; a build right after one of the same scene
    # must not (the hint held: no "ingest" timer)
    assert used_box_pass[1] is False and used_box_pass[2] is False
    # with the hint disabled every build runs the box pass and still agrees
    set_option("NO_GEOM_HINT", 1)
    pts = scenes["same_box"]
    d = _device_cloud(ctx, pts)
    g = Forest(0, np.zeros(3), 1.0)
    g.add_pose_device(d, len(pts), adopt=True)
    g.subdivide(K)
    _assert_same_tables(_tables(g), want["same_box"])
    g.close()
    ctx.check(ctx.lib.octl_dev_free(ctx.handle, d))


def test_adopted_store_becomes_the_forests_own_when_it_grows():
    """More poses after an adopted one, and points appended to the adopted pose: the store is copied into the
    forest's own block first; the caller's buffer is never written."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    ctx = nat.get_context()
    rng = np.random.default_rng(77)
    a, b, extra = rng.random((5001, 3)) * 4, rng.random((3000, 3)) * 4 - 1.0, rng.random((777, 3)) * 4
    d = _device_cloud(ctx, a)
    g = Forest(0, np.zeros(3), 1.0)
    g.add_pose_device(d, len(a), adopt=True)
    g.add_pose(b)                      # before any build: the pending box is folded in, the store copied
    g.subdivide(20)
    h = Forest(0, np.zeros(3), 1.0)
    h.add_pose(a)
    h.add_pose(b)
    h.subdivide(20)
    _assert_same_tables(_tables(g), _tables(h))
    g.extend_pose(0, extra)
    h.extend_pose(0, extra)
    g.build(0, keep_scheme=True)
    h.build(0, keep_scheme=True)
    g.subdivide(10)
    h.subdivide(10)
    _assert_same_tables(_tables(g), _tables(h))
    back = np.empty_like(a)
    ctx.check(ctx.lib.octl_dev_download(ctx.handle, nat.ptr(back), d, a.nbytes))
    assert np.array_equal(back, a)
    g.close()
    h.close()
    # adopt, build, then append to the adopted pose directly
    g = Forest(0, np.zeros(3), 1.0)
    g.add_pose_device(d, len(a), adopt=True)
    g.subdivide(20)
    g.extend_pose(0, extra)
    g.subdivide(20)
    h = Forest(0, np.zeros(3), 1.0)   # (the same history: the epochs of the internal nodes record it)
    h.add_pose(a)
    h.subdivide(20)
    h.extend_pose(0, extra)
    h.subdivide(20)
    _assert_same_tables(_tables(g), _tables(h))
    assert np.array_equal(np.sort(g.xyz, axis=0), np.sort(np.vstack([a, extra]), axis=0))
    g.close()
    h.close()
    ctx.check(ctx.lib.octl_dev_download(ctx.handle, nat.ptr(back), d, a.nbytes))
    assert np.array_equal(back, a)
    ctx.check(ctx.lib.octl_dev_free(ctx.handle, d))


def test_async_host_feed_builds_the_same_forest_and_overlaps_nothing_it_should_not():
    """octreelib_amd.upload_async + Grid.insert_points(DeviceCloud): scans uploaded on the copy stream from
    page-locked AND from pageable host memory, consumed in and out of order, must build exactly what the
    synchronous path builds - first pose read in place, later poses copied on the device."""
    import octreelib_amd as oa
    from octreelib_amd import synthetic
    from octreelib_amd.grid import Grid, GridConfig

    scans = [synthetic.planar_cloud(150_000, (6, 6, 6), seed=1, stream=s) for s in range(4)]
    K = 40

    def reference(clouds):
        g = Grid(GridConfig(voxel_edge_length=1))
        for p, c in enumerate(clouds):
            g.insert_points(p, c)
        g.subdivide([oa.MaxPoints(K)])
        np.random.seed(5)
        g.map_leaf_points_cuda_ransac(hypotheses_number=128)
        t = _tables(g._forest)
        g._forest.close()
        return t

    want = [reference([s]) for s in scans]
    pinned = [oa.pinned_empty((len(s), 3)) for s in scans]
    for p, s in zip(pinned, scans):
        p[:] = s
    # a loop over scans: upload i+1 while i is processed
    nxt = oa.upload_async(pinned[0])
    for i in range(4):
        cur = nxt
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, cur)
        nxt = oa.upload_async(pinned[i + 1] if i % 2 else scans[i + 1]) if i + 1 < 4 else None   # pinned / pageable
        g.subdivide([oa.MaxPoints(K)])
        np.random.seed(5)
        g.map_leaf_points_cuda_ransac(hypotheses_number=128)
        _assert_same_tables(_tables(g._forest), want[i])
        with pytest.raises(RuntimeError, match="reads this buffer in place"):
            cur.release()          # the grid still reads the buffer
        g._forest.close()
        cur.release()
    # all uploads first, consumed in reverse order; a second pose from a DeviceCloud (device-to-device copy)
    ups = [oa.upload_async(p) for p in pinned]
    for i in (3, 1):
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, ups[i])
        g.subdivide([oa.MaxPoints(K)])
        np.random.seed(5)
        g.map_leaf_points_cuda_ransac(hypotheses_number=128)
        _assert_same_tables(_tables(g._forest), want[i])
        g._forest.close()
    g = Grid(GridConfig(voxel_edge_length=1))
    g.insert_points(0, ups[0])
    g.insert_points(1, ups[2])
    g.subdivide([oa.MaxPoints(K)])
    np.random.seed(5)
    g.map_leaf_points_cuda_ransac(hypotheses_number=128)
    _assert_same_tables(_tables(g._forest), reference([scans[0], scans[2]]))
    g._forest.close()
    ups[0].wait()
    pinned[0][:] = 0.0     # the host buffer is the caller's again after wait()
    for u in ups:
        u.release()


# ------------------------------------------------------------------------------------------------
# map_leaf_points with functions that REPLACE the leaf's cloud (octree.py:114-123)
# ------------------------------------------------------------------------------------------------
def test_grid_map_leaf_points_transform_golden():
    """fewer rows (pose subset), more rows, rows that leave their cubes, then a count filter: the reference's
    leaf contents at every stage (tests/golden/grid_map_transform.npz)."""
    from octreelib_amd.grid import Grid, GridConfig
    from tests.test_oracle_golden import canon_rows, golden_rows, run_map_transform_sequence

    g = load_golden("grid_map_transform.npz")
    grid = Grid(GridConfig(voxel_edge_length=1))
    for p in range(2):
        grid.insert_points(p, g[f"points{p}"])
    grid.subdivide(crit(int(g["K"])))

    def snap(tag):
        for p in range(2):
            got = canon_rows([(v.corner_min, v.edge_length, v.get_points()) for v in grid.get_leaf_points(p)])
            assert got == golden_rows(g, f"{tag}_p{p}")
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    run_map_transform_sequence(grid, snap)
    # get_points returns the transformed clouds
    for p in range(2):
        rows = np.vstack([np.empty((0, 3))] + [v.get_points() for v in grid.get_leaf_points(p)])
        assert sorted(map(bytes, grid.get_points(p))) == sorted(map(bytes, rows))
    assert grid.n_points(0) == 0 and grid.n_points(1) > 0   # (pose 0's five-row leaves did not pass len > 6)
    # rows outside their cubes fail a later subdivide, as the reference's IndexError does
    with pytest.raises((IndexError, ValueError)):
        grid.subdivide(crit(3))


def test_map_leaf_points_transform_on_manager_and_octree_vs_oracle():
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp
    from tests.test_oracle_golden import canon_rows

    rng = np.random.default_rng(5)
    poses = [rng.random((900, 3)), rng.random((700, 3)), rng.random((20, 3))]
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p, c in enumerate(poses):
        m.insert_points(p, c)
        om.insert_points(p, c)
    m.subdivide(crit(30))
    om.subdivide(30)
    shrink = lambda pts: (pts + pts.min(axis=0)) * 0.5     # stays inside the leaf's cube: a later subdivide works
    halves = lambda pts: pts[pts[:, 0] >= np.median(pts[:, 0])] * 1.0 - 0.0   # a selection, handed back as new values
    for fn, sel in ((shrink, [0, 2]), (halves, None), (lambda pts: np.empty((0, 3)), [1])):
        m.map_leaf_points(fn, sel)
        om.map_leaf_points(fn, sel)
        for p in range(3):
            got = canon_rows([(v.corner_min, v.edge_length, v.get_points()) for v in m.get_leaf_points(True, p)])
            want = canon_rows([(v.corner, v.edge, om.octrees[p].points[v.idx]) for v in om.octrees[p].leaves()])
            assert got == want
            assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == [om.n_nodes(p), om.n_leaves(p), om.n_points(p)]
    # the transformed clouds are ordinary points afterwards: the scheme is rebuilt from them
    m.subdivide(crit(10))
    clouds = {p: om.octrees[p].get_points() for p in range(3)}
    om2 = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(3):
        om2.insert_points(p, clouds[p])
    om2.subdivide(10)
    for p in range(3):
        got = canon_rows([(v.corner_min, v.edge_length, v.get_points()) for v in m.get_leaf_points(True, p)])
        want = canon_rows([(v.corner, v.edge, om2.octrees[p].points[v.idx]) for v in om2.octrees[p].leaves()])
        assert sorted(got) == sorted(want)   # (the listing order records the history: om2 has a shorter one)


# ------------------------------------------------------------------------------------------------
# voxel indices far beyond 2^20 (UTM-like coordinates at 1 m voxels): the reference takes any int64
# (grid.py:72-76); the packed voxel keys are relative to where the scene started
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("general_path", [False, True])
def test_grid_at_utm_scale_coordinates_vs_oracle(monkeypatch, general_path):
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp
    from oracle import ransac_np as rnp

    if general_path:
        set_option("NO_BUCKET_BUILD", 1)
    rng = np.random.default_rng(123)
    origin = np.array([5_432_100.0, -4_321_000.0, 1_250_000.0])   # |voxel index| up to 5.4e6 >> 2^20
    poses = [origin + rng.random((6000, 3)) * 5.0, origin + rng.random((4000, 3)) * 5.0 + np.array([3.0, -2.0, 1.0]),
             origin + rng.random((3000, 3)) * 4.0 + np.array([-6.0, 7.0, -3.0])]
    grid = Grid(GridConfig(voxel_edge_length=1))
    og = onp.OGrid(1)
    for p in range(2):
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])
    grid.subdivide(crit(30))
    og.subdivide(30)

    def check(n_poses):
        for p in range(n_poses):
            index = index_map(poses[p])
            assert_same_leaves(canon_from_list(views_table(grid.get_leaf_points(p), index)),
                               canon_from_list(og.leaf_table(p)))
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]

    check(2)
    # a late pose with new voxels on both sides (incremental insertion), then a re-subdivide over the old scheme
    grid.insert_points(2, poses[2])
    og.insert_points(2, poses[2])
    check(3)
    # (all poses, smaller K: the new scheme is finer or equal everywhere - coarsening is outside the parity domain,
    #  SURVEY 8 a8, and would leave the merged leaves of the oracle in child-major instead of insertion order)
    grid.subdivide(crit(12))
    og.subdivide(12)
    check(3)
    v = grid.get_leaf_points(1)[0]
    assert np.abs(v.corner_min).max() > 1e6 and v.corner_min.dtype.kind in "if"
    # RANSAC + apply_mask + filter at these coordinates
    np.random.seed(9)
    table = np.random.random((256, 6))
    np.random.seed(9)
    grid.map_leaf_points_cuda_ransac(poses_per_batch=2, hypotheses_number=256, threshold=0.05)
    for batch in ((0, 1), (2,)):
        rows = [og.leaf_table(p) for p in batch]
        cloud = np.vstack([poses[p][i] for p, t in zip(batch, rows) for _, _, i in t])
        sizes = np.array([len(i) for t in rows for _, _, i in t], dtype=np.int32)
        mask = rnp.evaluate(cloud, sizes, table, 0.05)
        off = 0
        for p, t in zip(batch, rows):
            n = sum(len(i) for _, _, i in t)
            og.apply_mask(p, mask[off : off + n])
            off += n
    check(3)
    grid.filter([lambda pts: len(pts) >= 4])
    og.filter([lambda pts: len(pts) >= 4])
    check(3)
    # get_points: all managers in first-creation order
    for p in range(3):
        assert sorted(map(bytes, grid.get_points(p))) == sorted(map(bytes, np.vstack([poses[p][i] for _, _, i in og.leaf_table(p)] + [np.empty((0, 3))])))


def test_scene_that_moves_more_than_2_pow_20_voxels_fails_loudly():
    from octreelib_amd import _native as nat
    from octreelib_amd.grid import Grid, GridConfig

    rng = np.random.default_rng(1)
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, rng.random((500, 3)) * 3.0 + 7_000_000.0)
    grid.subdivide(crit(20))
    assert grid.n_points(0) == 500
    grid.insert_points(1, rng.random((500, 3)) * 3.0 + 7_000_000.0 + 900_000.0)   # within 2^20 voxels: fine
    assert grid.n_points(1) == 500
    grid.insert_points(2, rng.random((500, 3)) * 3.0 + 7_000_000.0 + 1_200_000.0)   # beyond
    with pytest.raises((nat.DomainError, IndexError, ValueError)):
        grid.n_points(2)
    with pytest.raises((nat.DomainError, IndexError, ValueError)):     # absolute index beyond int32 range
        g2 = Grid(GridConfig(voxel_edge_length=1))
        g2.insert_points(0, rng.random((10, 3)) + 3.0e9)
        g2.n_points(0)


def test_thin_buckets_go_to_the_level_loop_and_match_the_oracle():
    """A scene that is sparse in a LARGE box (here 400 k points in 2048 x 2048 x 8 voxels: more than 4096 key
    ranges of 2^12 voxel keys, ~50 points in each) is not the bucket build's case - a workgroup per bucket is
    mostly set-up there.  The level loop takes it, and the result is the oracle's, leaf by leaf."""
    from octreelib_amd import _native as nat, synthetic
    from octreelib_amd._engine import Forest
    from tests._fullsize import oracle_check_every_leaf

    cloud = synthetic.sparse_scene(400_000, (2048, 2048, 8), seed=3)
    poses = [cloud[:150_000], cloud[150_000:]]
    ctx = nat.get_context()
    for rounds in range(2):           # the second build knows the scene is sparse: no single-pass attempt
        f = Forest(0, np.zeros(3), 1.0)
        for c in poses:
            f.add_pose(c)
        ctx.set_profiling(True)
        f.subdivide(6)
        names = set(ctx.timings())
        ctx.set_profiling(False)
        assert "keygen" in names                      # the general path built the tables
        if rounds:
            assert "bucket_build" not in names and "part_hist" not in names
        oracle_check_every_leaf(f, poses, 6, grid=True)
        f.close()
    # the same through a cloud that is read in place (uploaded from page-locked memory, adopted by the forest)
    import octreelib_amd as oa
    stage = oa.pinned_empty(cloud.shape)
    stage[:] = cloud
    dev = oa.upload_async(stage)
    f = Forest(0, np.zeros(3), 1.0)
    f.add_pose(dev)
    assert f.reads_in_place(dev)
    ctx.set_profiling(True)
    f.subdivide(6)
    names = set(ctx.timings())
    ctx.set_profiling(False)
    assert "keygen" in names and "bucket_build" not in names
    oracle_check_every_leaf(f, [cloud], 6, grid=True)
    f.close()
    dev.release()
    # a dense scene afterwards: the bucket build again
    dense = np.random.default_rng(0).random((300_000, 3)) * np.array([24.0, 24.0, 24.0])
    f = Forest(0, np.zeros(3), 1.0)
    f.add_pose(dense)
    ctx.set_profiling(True)
    f.subdivide(6)
    names = set(ctx.timings())
    ctx.set_profiling(False)
    f.close()
    assert "bucket_build" in names and "keygen" not in names


# ------------------------------------------------------------------------------------------------
# the six child digits of a record at once (ref_arith.h: digits18_exact) against the level-by-level form
# ------------------------------------------------------------------------------------------------
def _digit_boundary_cloud(rng, n, L, lo, hi):
    """Uniform points plus points ON and one / two ulps beside the cube faces of the first seven levels."""
    base = rng.uniform(lo, hi, (n, 3))
    k = rng.integers(int(lo * 128 / L), int(hi * 128 / L), (n // 2, 3)).astype(np.float64) * (L / 128.0)
    edge = [k, np.nextafter(k, np.inf), np.nextafter(np.nextafter(k, np.inf), np.inf)]
    if lo >= 0:
        # (one ulp BELOW a face only for non-negative coordinates: a negative coordinate within ~2^-54 of a cube's
        #  upper face rounds p - corner to the full edge - the reference's IndexError, SURVEY 8a, a DomainError here)
        edge.append(np.nextafter(k, -np.inf))
    mix = [np.where(rng.random(k.shape) < 0.5, e, rng.uniform(lo, hi, k.shape)) for e in edge]
    pts = np.vstack([base] + mix)
    pts = pts[(pts >= lo).all(axis=1) & (pts < hi).all(axis=1)]
    return np.unique(pts, axis=0)


@pytest.mark.parametrize("L,lo,hi", [(1, 0.0, 6.0), (2, 0.0, 12.0), (5, 0.0, 20.0), (4, -8.0, 8.0), (1, 1000.0, 1004.0)])
@pytest.mark.parametrize("general_path", [False, True])
def test_six_digits_at_once_equal_the_level_by_level_form(monkeypatch, L, lo, hi, general_path):
    """Non-negative coordinates under an integer cube: every rounded subtraction p - corner of the reference's
    descent (octree.py:73-75) is exact, so the six child digits a partition record carries are the leading bits of
    p - corner (one multiply and one conversion per axis for a power-of-two edge, the walk on the exact remainder
    for L = 5).  The builds with the short form switched off (OCTL_NO_EXACT_DIGITS: the reference's own operation
    order, level by level) must give the same tables bit for bit - on both build paths, with points on and one
    ulp beside the faces of every level, and with negative coordinates in the cloud (they keep the long form)."""
    from octreelib_amd._engine import Forest

    if general_path:
        set_option("NO_BUCKET_BUILD", 1)
    rng = np.random.default_rng(int(L * 100 + hi))
    pts = _digit_boundary_cloud(rng, 60_000, float(L), lo, hi)

    def tables(K):
        f = Forest(0, np.zeros(3), float(L))
        f.add_pose(pts)
        f.subdivide(K)
        t = _tables(f)
        f.close()
        return t

    for K in (40, 3):
        fast = tables(K)
        set_option("NO_EXACT_DIGITS", 1)
        slow = tables(K)
        set_option("NO_EXACT_DIGITS", 0)
        _assert_same_tables(fast, slow)
        assert int(fast[0]["depth"].max()) >= (2 if K == 40 else 4)


def test_six_digits_at_once_in_a_single_cube(monkeypatch):
    """The same for a bare Octree over an integer cube (more than 65 535 points: the level loop's fused level-0
    pass) and for a cube at a fractional corner, which keeps the long form whatever the switch says."""
    from octreelib_amd.octree import Octree, OctreeConfig

    rng = np.random.default_rng(77)
    for corner, edge in ((np.array([3.0, 0.0, 7.0]), 2.0), (np.array([0.25, 0.5, 0.125]), 1.0)):
        pts = corner + _digit_boundary_cloud(rng, 70_000, edge, 0.0, edge)
        pts = pts[((pts >= corner) & (pts < corner + edge)).all(axis=1)]
        tabs = []
        for off in (False, True):
            if off:
                set_option("NO_EXACT_DIGITS", 1)
            oc = Octree(OctreeConfig(), corner, np.float64(edge))
            oc.insert_points(pts)
            oc.subdivide(crit(50))
            tabs.append(_tables(oc._forest))
            if off:
                set_option("NO_EXACT_DIGITS", 0)
        _assert_same_tables(tabs[0], tabs[1])
        assert len(pts) > 65_535 and int(tabs[0][0]["depth"].max()) >= 3


# ------------------------------------------------------------------------------------------------
# a big single cube: the store partitioned once by its first levels, the level loop started below them
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["unit", "offset", "fractional", "negative", "uneven", "outside"])
def test_single_cube_prefix_partition_equals_the_plain_level_loop(monkeypatch, case):
    """BASELINE config 4's build path (build.hip: prefix partition + complete top tree + level loop from level pm),
    forced onto small clouds: node table, block table, permutation and listing order must be those of the plain
    level loop (itself pinned to the oracle and the reference's golden vectors) - for an integer cube (the exact
    digit form), a cube at an offset, a fractional and a negative corner (level-by-level digits), an uneven cloud
    whose top nodes do not all split (the path must step aside), and a point outside the cube (DomainError from
    the plain path, as before)."""
    from octreelib_amd import _native as nat
    from octreelib_amd._engine import Forest

    rng = np.random.default_rng(123)
    corner, edge = {"unit": ([0.0, 0.0, 0.0], 1.0), "offset": ([4.0, 8.0, 2.0], 2.0),
                    "fractional": ([0.3, -0.7, 2.25], 1.5), "negative": ([-3.0, -1.0, -2.0], 1.0),
                    "uneven": ([0.0, 0.0, 0.0], 1.0), "outside": ([0.0, 0.0, 0.0], 1.0)}[case]
    corner = np.array(corner)
    n = 300_000
    pts = corner + rng.random((n, 3)) * edge
    if case == "uneven":       # nearly everything in one octant: the other depth-1 nodes hold too little to split
        pts = corner + np.vstack([rng.random((n - 40, 3)) * 0.5, 0.5 + rng.random((40, 3)) * 0.5]) * edge
    if case == "outside":
        pts[1234] = corner + np.array([1.5, 0.2, 0.2]) * edge
    pts = pts[((pts >= corner) & (pts < corner + edge)).all(axis=1) | (case == "outside")]
    ctx = nat.get_context()

    def build(K, plain):
        if plain:
            set_option("NO_CUBE_PREFIX", 1)
        else:
            set_option("NO_CUBE_PREFIX", 0)
        f = Forest(1, corner, edge)
        f.add_pose(pts[: len(pts) // 2])
        f.add_pose(pts[len(pts) // 2 :])         # two poses: (leaf, pose) blocks need the ORIGINAL indices
        ctx.set_profiling(True)
        try:
            f.subdivide(K)
            names = set(ctx.timings())
            t = _tables(f)
        finally:
            ctx.set_profiling(False)
            f.close()
        return t, names

    set_option("CUBE_PREFIX_MIN", 100000)
    for K, levels in ((30, 4), (200, 3), (2000, 2)):
        if case == "outside":
            for plain in (False, True):
                with pytest.raises((IndexError, ValueError)):
                    build(K, plain)
            continue
        got, names = build(K, False)
        want, plain_names = build(K, True)
        _assert_same_tables(got, want)
        assert "prefix_scatter" not in plain_names
        # (300 000 points: >= 2 K points per depth-pm node for pm = 4 / 3 / 2; the uneven cloud makes the path step aside)
        assert "level_hist" in names
        if case != "uneven":
            assert "prefix_scatter" in names and "keygen" not in names
            assert int(got[0]["depth"].max()) >= levels


def _step_tables(pts_list, K, adopt_ctx=None, H=256):
    """insert (one pose per cloud) + subdivide + RANSAC + apply_mask through the engine; the tables before and after
    the mask."""
    from octreelib_amd._engine import Forest

    f = Forest(0, np.zeros(3), 1.0)
    bufs = []
    for pts in pts_list:
        if adopt_ctx is not None and not bufs:
            d = _device_cloud(adopt_ctx, pts)
            bufs.append(d)
            f.add_pose_device(d, len(pts), adopt=True)
        else:
            f.add_pose(pts)
    f.subdivide(K)
    before = _tables(f)
    np.random.seed(0)
    table = np.random.random((H, 6))
    f.ransac_all(10, table, 0.01)
    f.apply_device_mask()
    after = (({k: v.copy() for k, v in f.blocks.items()}), f.perm.copy(), f.xyz.copy())
    f.close()
    for d in bufs:
        adopt_ctx.check(adopt_ctx.lib.octl_dev_free(adopt_ctx.handle, d))
    return before, after


def _assert_same_step(a, b, canonical=False):
    if canonical:
        # (bucket path + level loop for the voxels it left behind numbers those voxels' nodes behind the others;
        #  the level-synchronous path numbers everything level-major: compare independent of the numbering)
        na, ba = _canon_build(a[0])
        nb, bb = _canon_build(b[0])
        assert na == nb and ba == bb
        for x, y in zip(a[0][2:], b[0][2:]):
            assert np.array_equal(x, y)
    else:
        _assert_same_tables(a[0], b[0])
    assert a[1][0].keys() == b[1][0].keys()
    for k in a[1][0]:
        if canonical and k == "node":
            continue   # (node ids depend on the numbering; the blocks' order, slots, starts and sizes do not)
        assert np.array_equal(a[1][0][k], b[1][0][k]), k
    assert np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[1][2], b[1][2])


@pytest.mark.parametrize("scene", ["even", "tiny", "skewed", "two_poses", "two_pass"])
def test_round5_launch_trimming_changes_no_result(scene):
    """Round 5 took launches out of the step - the partition table in one kernel (k_table_scan), the bucket totals'
    scan with a last-workgroup epilogue, RANSAC preparation and apply_mask as look-back kernels, chunk kernels and
    RANSAC instances only when needed, host waits that poll the pinned mirror.  Every one of them has a switch back to
    the round-4 form (octl_debug_set_option): the results must not depend on any of them - scheme, blocks, order,
    permutation before the mask; blocks, permutation and coordinates after it."""
    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic

    ctx = nat.get_context()
    K = 40
    if scene == "even":
        clouds = [synthetic.planar_cloud(150_000, (6, 6, 6), seed=1, stream=3)]
    elif scene == "tiny":
        clouds = [synthetic.planar_cloud(900, (2, 2, 2), seed=1, stream=3)]
    elif scene == "skewed":   # buckets beyond 4096 points: the chunk kernels
        clouds = [synthetic.sparse_scene(200_000, (24, 24, 8), seed=5, cluster_fraction=0.3, cluster_density=60.0)]
    elif scene == "two_poses":
        clouds = [synthetic.planar_cloud(60_000, (5, 5, 5), seed=1, stream=1),
                  synthetic.planar_cloud(50_000, (5, 5, 5), seed=1, stream=2)]
    else:
        clouds = [synthetic.planar_cloud(120_000, (16, 16, 16), seed=1, stream=4)]
        set_option("BUCKET_POINTS", 16)   # > 4096 buckets: the two-pass partition
    adopt = ctx if scene != "two_poses" else None
    want = _step_tables(clouds, K, adopt)
    again = _step_tables(clouds, K, adopt)          # (second build of the context: hinted geometry, chunk history)
    _assert_same_step(want, again)
    for opt in ("NO_FUSED_TABLES", "NO_SPIN_WAIT", "NO_SPEC_FINISH", "GEOM_MARGIN", "NO_GEOM_HINT", "NO_BUCKET_BUILD"):
        set_option(opt, 1 if opt != "GEOM_MARGIN" else -1)
        got = _step_tables(clouds, K, adopt)
        set_option(opt, 0)
        _assert_same_step(want, got, canonical=(opt == "NO_BUCKET_BUILD"))
        _assert_same_step(want, _step_tables(clouds, K, adopt))   # ... and back


def test_small_scans_through_the_bucket_path_equal_the_level_loop():
    """A scan of a few points up to a LiDAR sweep: the single-pass tables are sized by the bucket count the cloud asks
    for (64 ... 4096 buckets), whatever that is the bucket path must build the level loop's tables."""
    from octreelib_amd import synthetic
    from octreelib_amd._engine import Forest

    for n in (1, 2, 7, 63, 64, 65, 200, 3000, 30_000, 100_000, 300_000):
        pts = synthetic.planar_cloud(n, (7, 7, 7), seed=1, stream=n)
        tabs = []
        for general in (0, 1):
            set_option("NO_BUCKET_BUILD", general)
            f = Forest(0, np.zeros(3), 1.0)
            f.add_pose(pts)
            f.subdivide(20)
            tabs.append(_tables(f))
            f.close()
        set_option("NO_BUCKET_BUILD", 0)
        _assert_same_tables(tabs[0], tabs[1])


@pytest.mark.gpu
def test_speculative_bucket_finish_holds_on_a_repeated_scan_and_misses_safely():
    """Round 5: from a context's second bucket build on, k_bucket_finish is enqueued BEFORE the host has seen the build's
    totals (tables sized from the previous build; the kernel checks on the device what the host checks in the mirror).
    A repeated scan must be served by the speculative launch; a scan that outgrows the previous one's tables, or leaves
    work to other paths, must be caught by the check and built the ordinary way - same scheme either way."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic

    ctx = nat.get_context()
    lib = ctx.lib

    def counters():
        h, m = C.c_uint64(0), C.c_uint64(0)
        ctx.check(lib.octl_debug_spec_finish(C.byref(h), C.byref(m)))
        return h.value, m.value

    small = synthetic.planar_cloud(40_000, (4, 4, 4), seed=3, stream=1)
    again = synthetic.planar_cloud(40_000, (4, 4, 4), seed=3, stream=2)
    big = synthetic.planar_cloud(400_000, (8, 8, 8), seed=3, stream=3)        # ten times the points, nodes and blocks
    spread = synthetic.planar_cloud(40_000, (14, 14, 14), seed=3, stream=4)   # the small scan's points in 43 x the voxels
    skew = synthetic.sparse_scene(200_000, (24, 24, 8), seed=5, cluster_fraction=0.3, cluster_density=60.0)
    K = 24
    want = {}
    set_option("NO_SPEC_FINISH", 1)
    for name, c in (("small", small), ("again", again), ("big", big), ("spread", spread), ("skew", skew)):
        want[name] = _step_tables([c], K, ctx)
    set_option("NO_SPEC_FINISH", 0)
    _step_tables([small], K, ctx)                    # (whatever ran before: the context's sizes are this scan's now)
    h0, m0 = counters()
    got = _step_tables([again], K, ctx)
    h1, m1 = counters()
    assert (h1 - h0, m1 - m0) == (1, 0), (h1 - h0, m1 - m0)
    _assert_same_step(want["again"], got)
    got = _step_tables([big], K, ctx)                # (tables in proportion to the points: may hold, may miss)
    h2, m2 = counters()
    assert (h2 - h1) + (m2 - m1) >= 1
    _assert_same_step(want["big"], got)
    _step_tables([small], K, ctx)
    h2, m2 = counters()
    got = _step_tables([spread], K, ctx)             # outgrows the tables sized from the small scan: caught on the device
    h3, m3 = counters()
    assert h3 - h2 == 0 and m3 - m2 >= 1, (h3 - h2, m3 - m2)
    _assert_same_step(want["spread"], got)
    got = _step_tables([skew], K, ctx)               # buckets beyond 4096 points (chunk kernels), a new voxel box
    _assert_same_step(want["skew"], got)
    got = _step_tables([skew], K, ctx)
    _assert_same_step(want["skew"], got)
    got = _step_tables([small], K, ctx)              # and back
    _assert_same_step(want["small"], got)


@pytest.mark.gpu
def test_a_scene_that_drifts_by_a_voxel_keeps_its_geometry_and_a_jump_costs_one_attempt():
    """Round 6: the key geometry is formed over the TRUE voxel box padded by a margin (one voxel), buckets are runs of
    any width, and the next build's hint is formed from the box the histogram pass finds on its way.  A scene whose
    box drifts by a voxel per scan never pays a box pass nor a rejected attempt; a jump beyond the margin costs ONE
    rejected attempt (the partition kernels launched twice) and the scan after it runs under a hint again.  Counted by
    the launches of a build; the results never depend on any of it (grid.py:72-90 rebuckets from scratch per call)."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic

    ctx = nat.get_context()
    lib = ctx.lib
    a = synthetic.planar_cloud(60_000, (5, 5, 5), seed=4, stream=1)
    a2 = synthetic.planar_cloud(60_000, (5, 5, 5), seed=4, stream=2)
    K = 32

    def moved(pts, dx, dy=0.0, dz=0.0):
        return pts + np.array([dx, dy, dz])

    def launches_of(pts):
        l0, l1 = C.c_uint64(0), C.c_uint64(0)
        ctx.check(lib.octl_debug_launches(C.byref(l0)))
        got = _step_tables([pts], K, ctx)
        ctx.check(lib.octl_debug_launches(C.byref(l1)))
        return l1.value - l0.value, got

    def truth(pts):
        set_option("NO_GEOM_HINT", 1)
        try:
            return _step_tables([pts], K, ctx)
        finally:
            set_option("NO_GEOM_HINT", 0)

    set_option("NO_GEOM_HINT", 1)
    nohint, _ = launches_of(a)                # every build with its own box pass
    set_option("NO_GEOM_HINT", 0)
    _step_tables([a], K, ctx)
    base, got = launches_of(a2)               # a build under a hint that holds
    _assert_same_step(truth(a2), got)
    _step_tables([a2], K, ctx)                # (truth() ran without a hint: one build to have one again)
    assert base < nohint, (base, nohint)
    # the scene walks away one voxel per scan, along x, then diagonally, then back: every build under a hint
    walk = [(1, 0, 0), (2, 0, 0), (3, 1, 0), (4, 2, 1), (3, 1, 0), (2, 0, -1), (1, -1, -2), (0, 0, -1)]
    for i, (dx, dy, dz) in enumerate(walk):
        pts = moved(a if i % 2 else a2, float(dx), float(dy), float(dz))
        n, got = launches_of(pts)
        assert n == base, (i, base, n)
        want = truth(pts)
        _assert_same_step(want, got)
        _step_tables([pts], K, ctx)           # (the hint again, formed from this cloud's box)
    # a jump of six voxels: the hint is rejected once, the build runs again from the box the attempt has found
    far = moved(a, 6.0, 0.0, -1.0)
    _step_tables([a], K, ctx)
    rej, got = launches_of(far)
    _assert_same_step(truth(far), got)
    assert rej > base + 3, (base, rej)
    _step_tables([far], K, ctx)
    n, got = launches_of(moved(a2, 7.0, 1.0, -1.0))   # ... and the next scan is followed again
    assert n == base, (base, n)
    # without the margin (the tight box of rounds 2-5) a one-voxel drift is a rejected attempt
    set_option("GEOM_MARGIN", -1)
    try:
        _step_tables([a], K, ctx)
        n0, got = launches_of(a2)
        _assert_same_step(truth(a2), got)
        _step_tables([a], K, ctx)
        n1, got = launches_of(moved(a2, 1.0))
        assert n1 > n0 + 3, (n0, n1)
    finally:
        set_option("GEOM_MARGIN", 0)
    # a wider margin follows a faster scene
    set_option("GEOM_MARGIN", 2)
    try:
        _step_tables([a], K, ctx)             # (under the one-voxel hint; the next hint has the wider margin)
        _step_tables([a], K, ctx)
        n2, got = launches_of(moved(a2, 2.0, -2.0, 1.0))
        _assert_same_step(truth(moved(a2, 2.0, -2.0, 1.0)), got)
        assert n2 == base, (base, n2)
    finally:
        set_option("GEOM_MARGIN", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["planar", "noise", "collinear", "clusters", "lattice", "two_planes", "thin", "far", "small_extent",
                                  "large_extent", "tight_threshold", "wide_threshold"])
def test_ransac_prescreen_never_changes_a_result(kind):
    """Round 6: after the first 64 hypotheses of a leaf the kernel fits a hypothesis exactly only when an approximate f32
    plane of its six sample points, with a rigorous bound on its distances, cannot rule out that it beats the best count
    so far (ransac.hip: prescreen_constants).  Whatever the prescreen does, count, winner index, f32 plane bits and mask
    must be the reference's (oracle) - on leaves where it prunes nearly everything (planar), nearly nothing (noise),
    where its bound is weak (collinear samples, point clusters: tiny cofactors), where counts tie all over (lattice),
    where a later hypothesis is the winner by one point (two planes), and where it must stand aside (coordinates,
    extents and thresholds outside its range).  The same launch with the prescreen switched off must agree too."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    import zlib

    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    sizes = rng.integers(6, 64, 1500).astype(np.int32)
    sizes[::50] = 63
    sizes[1::50] = 6
    n = int(sizes.sum())
    starts = np.concatenate(([0], np.cumsum(sizes)))
    base = rng.random((len(sizes), 3)) * 30.0          # one voxel-sized neighbourhood per block
    cloud = np.repeat(base, sizes, axis=0) + rng.random((n, 3)) * 0.5
    thr = 0.01

    def plane_z(s, e, a, b, sigma):
        cloud[s:e, 2] = base_of(s)[2] + 0.25 + a * (cloud[s:e, 0] - base_of(s)[0]) + b * (cloud[s:e, 1] - base_of(s)[1]) \
            + rng.normal(0, sigma, e - s)

    def base_of(s):
        return base[np.searchsorted(starts, s, side="right") - 1]

    for bi in range(len(sizes)):
        s, e = starts[bi], starts[bi + 1]
        if kind == "planar" or kind in ("far", "small_extent", "large_extent", "tight_threshold", "wide_threshold"):
            plane_z(s, e, rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), 0.005)
            out = rng.random(e - s) < 0.2
            cloud[s:e, 2][out] = base_of(s)[2] + rng.random(int(out.sum())) * 0.5
        elif kind == "collinear":     # points along a line + a little noise: every sample is nearly collinear
            t = rng.random(e - s)
            d = rng.normal(size=3)
            cloud[s:e] = base_of(s) + 0.25 + np.outer(t - 0.5, d / np.linalg.norm(d)) * 0.4 + rng.normal(0, 0.004, (e - s, 3))
        elif kind == "clusters":      # three tight clusters: samples within one cluster have tiny covariances
            c = rng.random((3, 3)) * 0.4
            which = rng.integers(0, 3, e - s)
            cloud[s:e] = base_of(s) + c[which] + rng.normal(0, 1e-4, (e - s, 3))
        elif kind == "lattice":       # counts tie everywhere; distances exactly on the threshold
            cloud[s:e] = np.floor(base_of(s)) + rng.integers(0, 8, (e - s, 3)) / 64.0
            cloud[s:e, 2] = np.floor(base_of(s)[2]) + rng.integers(0, 3, e - s) / 64.0
        elif kind == "two_planes":    # two parallel planes 2.5 thresholds apart, nearly equal populations
            plane_z(s, e, 0.1, -0.2, 0.002)
            cloud[s:e, 2] += 0.025 * (rng.random(e - s) < 0.5)
        elif kind == "thin":          # a sliver: extent along one axis only
            cloud[s:e, 1] = base_of(s)[1] + rng.random(e - s) * 1e-3
            plane_z(s, e, 0.3, 0.0, 0.004)
    if kind == "far":
        cloud += 3.0e5                 # the f32 rounding of the reference's global plane dwarfs the threshold
    elif kind == "small_extent":
        cloud = base.repeat(sizes, axis=0) + (cloud - base.repeat(sizes, axis=0)) * 2.0 ** -9
        thr = 0.01 * 2.0 ** -9
    elif kind == "large_extent":
        cloud = cloud * 600.0
        thr = 6.0
    elif kind == "tight_threshold":
        thr = 1.0e-5
    elif kind == "wide_threshold":
        thr = 0.3
    if kind != "lattice":
        assert len(np.unique(cloud, axis=0)) == n
    np.random.seed(11)
    op = CudaRansac(threshold=thr, hypotheses_number=1024, initial_points_number=6)
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, thr, details=True)
    for off in (0, 1):
        set_option("NO_RANSAC_PRESCREEN", off)
        mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
        assert np.array_equal(counts, o_count), off
        assert np.array_equal(index, o_index), off
        assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32)), off
        assert np.array_equal(mask, o_mask), off
    set_option("NO_RANSAC_PRESCREEN", 0)
    if kind in ("planar", "two_planes", "noise"):
        assert (o_index >= 64).any()     # winners beyond the first group exist: the survivors' path decides them


@pytest.mark.gpu
@pytest.mark.parametrize("n_blocks", [1, 257, 2048, 2049, 4097, 6145, 8192, 8193])
@pytest.mark.parametrize("H,k", [(1024, 6), (64, 3)])
def test_one_launch_preparation_of_a_small_ransac_launch(n_blocks, H, k):
    """Round 6: a launch of up to 8192 blocks is prepared by ONE kernel (k_block_prepare_small: descriptors, size
    classes, sorted list, position table), a larger one by k_block_prepare + k_block_scatter; NO_FUSED_TABLES takes the
    round-3 form (sizes, scan, descriptors, scatter).  All three must give the same count, winner, plane bits and mask
    for every block - blocks below k points (finished at once: cuda_ransac.py:96-97), up to 63, 64 .. 255 and beyond
    the register path's capacity in one batch - and the oracle's on a prefix of the batch."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(n_blocks * 7 + H + k)
    sizes = rng.integers(1, 64, n_blocks).astype(np.int32)
    sizes[::97] = rng.integers(64, 256, len(sizes[::97]))
    sizes[5::1500] = rng.integers(256, 700, len(sizes[5::1500]))
    sizes[3::11] = rng.integers(1, k, len(sizes[3::11]))     # finished at once
    n = int(sizes.sum())
    base = rng.random((n_blocks, 3)) * 40.0
    cloud = np.repeat(base, sizes, axis=0) + rng.random((n, 3)) * 0.5
    cloud[:, 2] = np.repeat(base[:, 2], sizes) + 0.2 * (cloud[:, 0] - np.repeat(base[:, 0], sizes)) \
        + rng.normal(0, 0.004, n)
    np.random.seed(5)
    op = CudaRansac(threshold=0.01, hypotheses_number=H, initial_points_number=k)
    res = {}
    try:
        for fused_off in (0, 1):
            set_option("NO_FUSED_TABLES", fused_off)
            res[fused_off] = op.evaluate(cloud, sizes, details=True)
    finally:
        set_option("NO_FUSED_TABLES", 0)
    (m0, p0, c0, i0), (m1, p1, c1, i1) = res[0], res[1]
    assert np.array_equal(c0, c1) and np.array_equal(i0, i1)
    assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32)) and np.array_equal(m0, m1)
    assert (c0[sizes < k] == 0).all() and (i0[sizes < k] == -1).all()
    # the oracle on the first blocks (a block's result depends on its own points and the point behind it only)
    nb = min(n_blocks, 120 if H == 1024 else 600)
    npts = int(sizes[:nb].sum())
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud[:npts + (1 if npts < n else 0)], sizes[:nb],
                                                        op.random_hypotheses, 0.01, details=True)
    assert np.array_equal(c0[:nb], o_count) and np.array_equal(i0[:nb], o_index)
    assert np.array_equal(p0[:nb].view(np.uint32), o_plane.view(np.uint32))
    assert np.array_equal(m0[:npts], o_mask[:npts])


@pytest.mark.gpu
def test_asynchronous_apply_mask_books_its_counts_when_somebody_asks():
    """Round 6: octl_forest_apply_mask_async returns behind its last launch; the surviving point / block counts are
    booked by the next call that looks at the forest (forest_settle at every entry point), dropped unread by clear /
    destroy, and a second compaction on the context - this forest or another - books the first before it starts
    (they share two words of the pinned mirror).  Whatever the order of calls, the tables are those of the waiting
    form (octl_forest_apply_mask; Octree.apply_mask, octree/octree.py:265-274)."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic
    from octreelib_amd._engine import Forest

    ctx = nat.get_context()
    lib = ctx.lib
    a = synthetic.planar_cloud(60_000, (5, 5, 5), seed=9, stream=1)
    b = synthetic.planar_cloud(45_000, (4, 4, 4), seed=9, stream=2)
    np.random.seed(0)
    table = np.random.random((256, 6))

    def prepared(pts):
        f = Forest(0, np.zeros(3), 1.0)
        f.add_pose(pts)
        f.subdivide(32)
        f.ransac_all(10, table, 0.01)
        return f

    def after(f):
        return ({k: v.copy() for k, v in f.blocks.items()}, f.perm.copy(), f.xyz.copy())

    def same(x, y):
        assert x[0].keys() == y[0].keys()
        for k in x[0]:
            assert np.array_equal(x[0][k], y[0][k]), k
        assert np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])

    # the waiting form: the truth
    want, want_n = {}, {}
    for name, pts in (("a", a), ("b", b)):
        f = prepared(pts)
        n = C.c_int64(0)
        ctx.check(lib.octl_forest_apply_mask(f.handle, C.byref(n)))
        f.n_ord = n.value
        f._invalidate()
        want[name], want_n[name] = after(f), n.value
        f.close()
    assert 0 < want_n["a"] < len(a)
    # 1. asynchronous, then the explicit settle
    f = prepared(a)
    ctx.check(lib.octl_forest_apply_mask_async(f.handle))
    n = C.c_int64(-1)
    ctx.check(lib.octl_forest_settle(f.handle, C.byref(n)))
    assert n.value == want_n["a"]
    ctx.check(lib.octl_forest_settle(f.handle, C.byref(n)))     # (a no-op the second time)
    assert n.value == want_n["a"]
    f.n_ord = n.value
    f._invalidate()
    same(want["a"], after(f))
    f.close()
    # 2. asynchronous, and the next call that looks at the forest books the counts (the engine's lazy n_ord)
    f = prepared(a)
    f.apply_device_mask()
    same(want["a"], after(f))
    assert f.n_ord == want_n["a"]
    # ... a second RANSAC + compaction on what is left, asynchronous again
    f.ransac_all(10, table, 0.01)
    f.apply_device_mask()
    n2 = f.n_ord
    assert 0 < n2 <= want_n["a"]
    f.close()
    # 3. two forests of one context: the second compaction books the first one's counts before it starts
    fa, fb = prepared(a), prepared(b)
    fa.apply_device_mask()
    fb.apply_device_mask()
    same(want["b"], after(fb))
    same(want["a"], after(fa))
    assert (fa.n_ord, fb.n_ord) == (want_n["a"], want_n["b"])
    # 4. clear with the counts still in flight: dropped, the forest builds the next cloud as a fresh one does
    fa.ransac_all(10, table, 0.01)
    fa.apply_device_mask()
    ctx.check(lib.octl_forest_clear(fa.handle))
    n = C.c_int64(-1)
    ctx.check(lib.octl_forest_settle(fa.handle, C.byref(n)))
    assert n.value == 0
    fa.close()
    # 5. destroy with the counts in flight
    fb.ransac_all(10, table, 0.01)
    fb.apply_device_mask()
    fb.close()
    f = prepared(b)
    f.apply_device_mask()
    same(want["b"], after(f))
    f.close()


def _input_forms(pts64):
    """The same cloud as the reference's callers may hand it over (the reference upcasts everything to f64:
    internal/voxel.py:81-83, octree/octree.py:100): values are f32-representable so that every form holds the same
    numbers."""
    strided = np.zeros((len(pts64), 7), dtype=np.float64)
    strided[:, 1::2] = pts64
    wide = np.zeros((2 * len(pts64), 3), dtype=np.float64)
    wide[::2] = pts64
    return {
        "f64_c": np.ascontiguousarray(pts64),
        "f32": pts64.astype(np.float32),
        "f64_fortran": np.asfortranarray(pts64),
        "strided_columns": strided[:, 1::2],
        "strided_rows": wide[::2],
        "reversed_view": pts64[::-1][::-1],
        "list": pts64.tolist(),
        "f16_upcast_f32": pts64.astype(np.float32).astype(np.float64).astype(np.float32),
    }


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["f32", "f64_fortran", "strided_columns", "strided_rows", "reversed_view", "list", "f16_upcast_f32"])
def test_input_dtypes_and_layouts_through_the_drop_in_classes(form):
    """Grid / Octree / OctreeManager .insert_points and CudaRansac.evaluate take whatever array-like the reference's
    callers pass - f32, Fortran order, strided views, Python lists - and must give the results of the f64 C-contiguous
    cloud (which the oracle pins)."""
    from octreelib_amd import MaxPoints
    from octreelib_amd.grid import Grid, GridConfig
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from octreelib_amd.ransac import CudaRansac
    from oracle import octree_np as onp
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(3)
    pts = (rng.random((6000, 3)) * 4.0 - 1.0).astype(np.float32).astype(np.float64)   # (negatives too; f32-exact values)
    pts = np.unique(pts, axis=0)
    rng.shuffle(pts)
    forms = _input_forms(pts)
    base, alt = forms["f64_c"], forms[form]
    if form != "list":
        assert not (isinstance(alt, np.ndarray) and alt.dtype == np.float64 and alt.flags.c_contiguous and form != "reversed_view")
    K = 16

    def leaves_of(obj, getter):
        index = index_map(pts)
        return canon_from_list(views_table(getter(obj), index))

    # Grid
    tabs = []
    for cloud in (base, alt):
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, cloud)
        g.subdivide([MaxPoints(K)])
        tabs.append((leaves_of(g, lambda o: o.get_leaf_points(0)), g.n_nodes(0), g.n_leaves(0), g.n_points(0)))
        assert g.get_points(0).dtype == np.float64
    assert tabs[0][1:] == tabs[1][1:]
    assert_same_leaves(tabs[0][0], tabs[1][0])
    og = onp.OGrid(1)
    og.insert_points(0, base)
    og.subdivide(K)
    assert_same_leaves(tabs[1][0], _oracle_pose_table(og, 0))
    # Octree and OctreeManager over the cube [-1, 3)^3
    cube = pts
    otabs = []
    for cloud in (base, alt):
        o = Octree(OctreeConfig(), np.array([-1.0, -1.0, -1.0]), 4.0)
        o.insert_points(cloud)
        o.subdivide([MaxPoints(K)])
        m = OctreeManager(Octree, OctreeConfig(), np.array([-1.0, -1.0, -1.0]), 4.0)
        m.insert_points(0, cloud)
        m.subdivide([MaxPoints(K)])
        otabs.append((leaves_of(o, lambda x: x.get_leaf_points()), o.n_nodes, o.n_leaves, o.n_points,
                      leaves_of(m, lambda x: x.get_leaf_points(True, 0)), m.n_nodes(0), m.n_leaves(0)))
    assert otabs[0][1:4] == otabs[1][1:4] and otabs[0][5:] == otabs[1][5:]
    assert_same_leaves(otabs[0][0], otabs[1][0])
    assert_same_leaves(otabs[0][4], otabs[1][4])
    assert_same_leaves(otabs[0][0], otabs[0][4])
    ot = onp.OTree(np.array([-1.0, -1.0, -1.0]), np.float64(4))
    ot.insert_points(cube)
    ot.subdivide(K)
    assert_same_leaves(otabs[1][0], canon_from_list(onp.tree_leaf_table(ot)))
    assert otabs[1][1:4] == (ot.n_nodes, ot.n_leaves, ot.n_points)
    # CudaRansac.evaluate: cloud and block sizes in the same forms
    sizes = rng.integers(0, 70, 120).astype(np.int32)
    cloud = pts[: int(sizes.sum())]
    cforms = _input_forms(cloud)
    np.random.seed(5)
    op = CudaRansac(threshold=0.05, hypotheses_number=1024, initial_points_number=6)
    want = op.evaluate(cforms["f64_c"], sizes, details=True)
    size_forms = {"f32": sizes.astype(np.float32), "list": sizes.tolist(), "strided_rows": np.repeat(sizes, 2)[::2],
                  "f64_fortran": sizes.astype(np.int64)}
    got = op.evaluate(cforms[form], size_forms.get(form, sizes), details=True)
    for a, b in zip(want, got):
        assert np.array_equal(np.asarray(a).view(np.uint8), np.asarray(b).view(np.uint8))
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, 0.05, details=True)
    assert np.array_equal(got[0], o_mask) and np.array_equal(got[2], o_count) and np.array_equal(got[3], o_index)
    assert np.array_equal(got[1].view(np.uint32), o_plane.view(np.uint32))
