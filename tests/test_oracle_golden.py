"""The oracle (oracle/octree_np.py, oracle/ransac_np.py) against the golden vectors that
were produced by the reference itself (tests/golden/make_golden.py) and against the
hand-written known answers of the reference's own tests."""

import numpy as np
import pytest

from oracle import octree_np as onp
from oracle import ransac_np as rnp
from tests._util import assert_same_leaves, canon_from_list, golden_canon, load_golden


@pytest.mark.parametrize("n", [2000, 20000])
@pytest.mark.parametrize("k", [8, 32, 256])
def test_octree_uniform(n, k):
    g = load_golden(f"octree_uniform_{n}.npz")
    t = onp.OTree(np.array([0.0, 0.0, 0.0]), np.float64(1))
    t.insert_points(g["points"])
    t.subdivide(k)
    assert_same_leaves(canon_from_list(onp.tree_leaf_table(t)), golden_canon(g, f"k{k}"))
    assert [t.n_nodes, t.n_leaves, t.n_points] == list(g[f"k{k}_counts"])
    # full cached-leaf list incl. empty leaves, order included
    all_leaves = onp.tree_leaf_table(t, non_empty=False)
    assert np.array_equal(np.array([c for c, _, _ in all_leaves]), g[f"k{k}_all_corners"])
    assert np.array_equal(np.array([e for _, e, _ in all_leaves]), g[f"k{k}_all_edges"])


def test_grid_L1_mixed():
    g = load_golden("grid_L1_mixed.npz")
    og = onp.OGrid(1)
    og.insert_points(0, g["points"])
    assert_same_leaves(canon_from_list(og.leaf_table(0)), golden_canon(g, "pre"))
    assert [og.n_nodes(0), og.n_leaves(0), og.n_points(0)] == list(g["pre_counts"])
    og.subdivide(16)
    assert_same_leaves(canon_from_list(og.leaf_table(0)), golden_canon(g, "k16"))
    assert [og.n_nodes(0), og.n_leaves(0), og.n_points(0)] == list(g["k16_counts"])


def test_grid_L5_two_poses():
    g = load_golden("grid_L5_two_poses.npz")
    og = onp.OGrid(5)
    for p in range(2):
        og.insert_points(p, g[f"points{p}"])
    og.subdivide(24)
    for p in range(2):
        assert_same_leaves(canon_from_list(og.leaf_table(p)), golden_canon(g, f"p{p}"))
        assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"p{p}_counts"])
    og.subdivide(6, [1])
    for p in range(2):
        assert_same_leaves(canon_from_list(og.leaf_table(p)), golden_canon(g, f"r_p{p}"))
        assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"r_p{p}_counts"])


def test_grid_late_poses():
    """Poses inserted after a subdivide (known voxels, new voxels on both sides of the old ones, two
    late poses in a row), then a refinement: the reference's leaf tables at every stage."""
    g = load_golden("grid_late_poses.npz")
    og = onp.OGrid(2)

    def check(tag, n):
        for p in range(n):
            assert_same_leaves(canon_from_list(og.leaf_table(p)), golden_canon(g, f"{tag}_p{p}"))
            assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    og.insert_points(0, g["points0"])
    og.subdivide(60)
    og.insert_points(1, g["points1"])
    check("a", 2)
    og.insert_points(2, g["points2"])
    check("b", 3)
    og.insert_points(3, g["points3"])
    og.insert_points(4, g["points4"])
    check("c", 5)
    og.subdivide(25)
    check("d", 5)


def test_manager_four_poses():
    g = load_golden("manager_four_poses.npz")
    m = onp.OManager(np.array([0.0, 0.0, 0.0]), 2.0)
    for p in range(3):
        m.insert_points(p, g[f"points{p}"])
    m.subdivide(40, [0, 2])
    m.insert_points(3, g["points3"])
    for p in range(4):
        got = canon_from_list(onp.tree_leaf_table(m.octrees[p]))
        assert_same_leaves(got, golden_canon(g, f"p{p}"))
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == list(g[f"p{p}_counts"])
    m.subdivide(25)
    for p in range(4):
        got = canon_from_list(onp.tree_leaf_table(m.octrees[p]))
        assert_same_leaves(got, golden_canon(g, f"r_p{p}"))
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == list(g[f"r_p{p}_counts"])


@pytest.mark.parametrize("name", ["h64", "h1024", "h32k3"])
def test_ransac_operator(name):
    """Reference kernel source (run under the simulator stand-in) vs the restatement.
    The reference's winner among tied hypotheses is a race: its mask must equal the mask
    of one of the tied planes, and its popcount must equal the maximal inlier count."""
    g = load_golden(f"ransac_{name}.npz")
    cloud, sizes, hyp, thr = g["cloud"], g["block_sizes"], g["hypotheses"], float(g["threshold"])
    mask, best_count, best_plane, best_index, tied = rnp.evaluate(cloud, sizes, hyp, thr, details=True)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    k = hyp.shape[1]
    for b in range(len(sizes)):
        s, e = starts[b], starts[b + 1]
        ref = g["mask"][s:e]
        if sizes[b] < k:
            assert not ref.any() and not mask[s:e].any()
            continue
        assert int(ref.sum()) == int(best_count[b])
        blk = cloud[s:e]
        ok = False
        for pl in tied[b]:
            p = pl.astype(np.float64)
            d = np.abs(((p[0] * blk[:, 0] + p[1] * blk[:, 1]) + p[2] * blk[:, 2]) + p[3])
            if np.array_equal(d < thr, ref):
                ok = True
                break
        assert ok, f"block {b}: reference mask matches none of the tied planes"
    # the restatement's own choice (lowest tied index) is one of the legal outcomes too
    for b in range(len(sizes)):
        assert int(mask[starts[b] : starts[b + 1]].sum()) == int(best_count[b])


def _plane_mask(plane32, blk, thr):
    """measure_distance(plane, point) < threshold (util.py:22-24) for one f32 plane."""
    p = np.asarray(plane32, dtype=np.float32).astype(np.float64)
    return np.abs(((p[0] * blk[:, 0] + p[1] * blk[:, 1]) + p[2] * blk[:, 2]) + p[3]) < thr


@pytest.mark.parametrize("name", ["h1024", "h256"])
def test_ransac_bench_leaves_against_recorded_reference_planes(name):
    """Hundreds of blocks cut from the benchmark scene's own leaves, with the reference kernel's shared
    best_plane / max_inliers_number recorded per block (cuda_ransac.py:125-146; the upstream API only
    returns the mask): the restatement's maximal count must equal the reference's, the reference's
    plane must be one of the restatement's tied planes bit for bit, its mask must be the mask of
    that plane, and where the maximum is attained by a single plane the restatement's lowest-index
    plane must be the reference's (north_star: normals within 1e-5; here: the same f32 bits)."""
    g = load_golden(f"ransac_bench_leaves_{name}.npz")
    cloud, sizes, hyp, thr = g["cloud"], g["block_sizes"], g["hypotheses"], float(g["threshold"])
    ref_plane, ref_max = g["ref_plane"], g["ref_max_inliers"]
    mask, best_count, best_plane, best_index, tied = rnp.evaluate(cloud, sizes, hyp, thr, details=True)
    starts = np.concatenate(([0], np.cumsum(sizes)))
    k = hyp.shape[1]
    assert len(sizes) >= (300 if name == "h1024" else 64)
    singles = 0
    for b in range(len(sizes)):
        s, e = starts[b], starts[b + 1]
        ref = g["mask"][s:e]
        if sizes[b] < k:
            assert ref_max[b] == -1 and not ref.any() and not mask[s:e].any()
            continue
        assert int(ref_max[b]) == int(best_count[b]), f"block {b}"
        assert any(np.array_equal(ref_plane[b].view(np.uint32), t.view(np.uint32)) for t in tied[b]), \
            f"block {b}: the reference's plane is none of the tied planes"
        assert np.array_equal(_plane_mask(ref_plane[b], cloud[s:e], thr), ref), f"block {b}"
        assert int(ref.sum()) == int(ref_max[b])
        if len(tied[b]) == 1:
            singles += 1
            assert np.array_equal(best_plane[b].view(np.uint32), ref_plane[b].view(np.uint32))
            assert np.max(np.abs(best_plane[b][:3].astype(np.float64) - ref_plane[b][:3])) <= 1e-5
            assert np.array_equal(mask[s:e], ref)
    assert singles >= 5  # (most planar leaves are explained by several hypotheses: ties are the rule)


def test_ransac_degenerate_block_all_inliers():
    g = load_golden("ransac_h64.npz")
    sizes = g["block_sizes"]
    starts = np.concatenate(([0], np.cumsum(sizes)))
    b = 4  # seven identical points: zero-norm plane -> every point is an inlier
    assert g["mask"][starts[b] : starts[b + 1]].all()


def test_reference_known_answers_octree():
    # test/octree/test_octree.py:33-62
    pc = np.array([[0, 0, 1], [0, 0, 2], [0, 0, 3], [9, 9, 8], [9, 9, 9]], dtype=float)
    t = onp.OTree(np.array([0, 0, 0]), np.float64(10))
    t.insert_points(pc)
    assert (t.get_points() == pc).all()
    t.subdivide(2)
    assert t.n_leaves == 3 and t.n_points == 5
    assert len(t.cached) == 15  # test_octree.py:30


def test_reference_known_answers_multi_pose():
    # test/octree/test_multi_pose.py:45-68, 93-164
    def make():
        m = onp.OManager(np.array([0, 0, 0]), 5)
        m.insert_points(0, np.array([[0, 0, 1], [0, 0, 2], [0, 0, 3]], dtype=float))
        m.insert_points(1, np.array([[1, 0, 1], [4, 0, 2], [0, 2, 3]], dtype=float))
        return m

    m = make()
    m.subdivide(2, [0])
    assert [m.n_nodes(0), m.n_nodes(1)] == [9, 9]
    assert [m.n_leaves(0), m.n_leaves(1)] == [2, 3]
    leaves0 = {(tuple(v.corner), float(v.edge)) for v in m.octrees[0].leaves()}
    assert leaves0 == {((0, 0, 0), 2.5), ((0, 0, 2.5), 2.5)}
    m = make()
    m.subdivide(1, None)
    assert [m.n_nodes(0), m.n_nodes(1)] == [33, 33]
    assert [m.n_leaves(0), m.n_leaves(1)] == [3, 3]
    leaves0 = {(tuple(v.corner), float(v.edge)) for v in m.octrees[0].leaves()}
    assert leaves0 == {((0, 0, 0.625), 0.625), ((0, 0, 1.25), 1.25), ((0, 0, 2.5), 1.25)}
    leaves1 = {(tuple(v.corner), float(v.edge)) for v in m.octrees[1].leaves()}
    assert leaves1 == {((0.625, 0, 0.625), 0.625), ((0, 1.25, 2.5), 1.25), ((2.5, 0, 0), 2.5)}


def test_reference_known_answers_grid():
    # test/grid/test_grid.py:14-93
    def make():
        g = onp.OGrid(5)
        g.insert_points(0, np.array([[0, 0, 1], [0, 0, 2], [0, 0, 3], [9, 9, 8], [9, 9, 9]], dtype=float))
        g.insert_points(1, np.array([[1, 0, 1], [4, 0, 2], [0, 2, 3], [5, 9, 9], [9, 3, 8]], dtype=float))
        return g

    g = make()
    assert [g.n_leaves(0), g.n_leaves(1)] == [2, 3]
    assert [g.n_nodes(0), g.n_nodes(1)] == [2, 3]
    g.subdivide(2)
    assert [g.n_leaves(0), g.n_leaves(1)] == [4, 5]
    assert [g.n_nodes(0), g.n_nodes(1)] == [26, 27]
    assert [g.n_points(0), g.n_points(1)] == [5, 5]
    g = make()
    g.subdivide(3)
    assert [g.n_leaves(0), g.n_leaves(1)] == [3, 5]
    with pytest.raises(ValueError, match="Cannot insert points to existing pose 0"):
        g.insert_points(0, np.zeros((1, 3)))


@pytest.mark.parametrize("poses_per_batch", [1, 2])
def test_grid_ransac_batches(poses_per_batch):
    """Grid.map_leaf_points_cuda_ransac over batches of poses (grid.py:149-215) restated on the oracle:
    batches of consecutive poses, one evaluate() per batch, masks applied per pose - against the reference's
    own result for three poses on a subdivided grid."""
    g = load_golden("grid_ransac_batches.npz")
    og = onp.OGrid(int(g["L"]))
    poses = [g[f"points{p}"] for p in range(3)]
    for p in range(3):
        og.insert_points(p, poses[p])
    og.subdivide(int(g["K"]))
    np.random.seed(int(g["seed"]))
    table = np.random.random((128, 6))          # CudaRansac.__init__ (cuda_ransac.py:39-41)
    for i in range(0, 3, poses_per_batch):
        batch = list(range(i, min(i + poses_per_batch, 3)))
        clouds, sizes = [], []
        for p in batch:
            for _, _, idx in og.leaf_table(p):
                clouds.append(poses[p][idx])
                sizes.append(len(idx))
        mask = rnp.evaluate(np.vstack(clouds), np.array(sizes, dtype=np.int32), table, 0.01)
        off = 0
        for p in batch:
            m = og.n_points(p)
            og.apply_mask(p, mask[off:off + m])
            off += m
    tag = f"b{poses_per_batch}"
    for p in range(3):
        assert_same_leaves(canon_from_list(og.leaf_table(p)), golden_canon(g, f"{tag}_p{p}"))
        assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"{tag}_p{p}_counts"])


def test_grid_filter():
    """Grid.filter with point-count criteria on a subdivided two-pose grid, twice in a row: the reference's
    leaf tables after each filter."""
    g = load_golden("grid_filter.npz")
    og = onp.OGrid(1)
    for p in range(2):
        og.insert_points(p, g[f"points{p}"])
    og.subdivide(int(g["K"]))
    for tag, criteria in (("ge5", [lambda pts: len(pts) >= 5]),
                          ("in3to12", [lambda pts: len(pts) > 2, lambda pts: 12 >= len(pts)])):
        og.filter(criteria)
        for p in range(2):
            assert_same_leaves(canon_from_list(og.leaf_table(p)), golden_canon(g, f"{tag}_p{p}"))
            assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"{tag}_p{p}_counts"])


def test_manager_extend():
    """OctreeManager.insert_points into poses that already exist, then a finer subdivide."""
    g = load_golden("manager_extend.npz")
    m = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(3):
        m.insert_points(p, g[f"points{p}"])
    m.subdivide(25)
    for p in (0, 1):
        m.insert_points(p, g[f"extra{p}"])

    def check(tag):
        for p in range(3):
            assert_same_leaves(canon_from_list(onp.tree_leaf_table(m.octrees[p])), golden_canon(g, f"{tag}_p{p}"))
            assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    check("a")
    m.subdivide(10)
    check("b")


def test_octree_subdivide_as():
    """Octree.subdivide_as between stand-alone octrees: after a coarser history of its own, the structure of
    another octree, then of a finer one."""
    g = load_golden("octree_subdivide_as.npz")
    corner, edge = np.array([0.0, 0.0, 0.0]), np.float64(g["edge"])
    a, b, b2 = onp.OTree(corner, edge), onp.OTree(corner, edge), onp.OTree(corner, edge)
    a.insert_points(g["pa"])
    b.insert_points(g["pb"])
    b2.insert_points(g["pb"])
    b.subdivide(100)
    b2.subdivide(30)
    a.subdivide(900)
    for tag, other in (("as100", b), ("as30", b2)):
        a.subdivide_as(other)
        assert_same_leaves(canon_from_list(onp.tree_leaf_table(a)), golden_canon(g, tag))
        assert [a.n_nodes, a.n_leaves, a.n_points] == list(g[f"{tag}_counts"])


def _spread(points):
    return len(points) > 40 and float(points.max(axis=0).max() - points.min(axis=0).min()) > 0.3


def test_grid_callable_criterion():
    """An arbitrary callable as criterion (scheme from pose 0), then a count criterion over both poses."""
    g = load_golden("grid_callable.npz")
    og = onp.OGrid(1)
    for p in range(2):
        og.insert_points(p, g[f"points{p}"])

    def check(tag):
        for p in range(2):
            assert_same_leaves(canon_from_list(og.leaf_table(p)), golden_canon(g, f"{tag}_p{p}"))
            assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    og.subdivide([_spread], [0])
    check("spread")
    og.subdivide(15)
    check("k15")


# ------------------------------------------------------------------------------------------------
# the count-only, level-synchronous oracle used at the configs' full sizes (oracle/count_scheme_np.py)
# against the reference's golden vectors and against the recursive oracle
# ------------------------------------------------------------------------------------------------
def _count_scheme_leaf_rows(cs, pose, n_pose_points, root_ids=None):
    """non-empty leaves of one pose as canonical ((corner bytes, edge bytes), sorted indices), unordered"""
    lf = cs.leaf_of[pose]
    order = np.argsort(lf, kind="stable")
    ids, starts = np.unique(lf[order], return_index=True)
    ends = np.append(starts[1:], len(order))
    return {((cs.corner[i] + 0.0).tobytes(), np.float64(cs.edge[i]).tobytes()): tuple(sorted(order[a:b].tolist()))
            for i, a, b in zip(ids, starts, ends)}


@pytest.mark.parametrize("n", [2000, 20000])
@pytest.mark.parametrize("k", [8, 32, 256])
def test_count_scheme_octree_uniform_golden(n, k):
    from oracle import count_scheme_np as cnp

    g = load_golden(f"octree_uniform_{n}.npz")
    cs = cnp.count_scheme([g["points"]], np.zeros((1, 3)), np.float64(1), k)
    assert _count_scheme_leaf_rows(cs, 0, n) == dict(golden_canon(g, f"k{k}"))
    n_nodes, n_leaves, n_points = g[f"k{k}_counts"]
    assert len(cs.edge) == n_nodes and len(np.unique(cs.leaf_of[0])) == n_leaves
    # every leaf incl. the empty ones, as a set of (corner, edge)
    want = {(c.tobytes(), np.float64(e).tobytes()) for c, e in zip(g[f"k{k}_all_corners"], g[f"k{k}_all_edges"])}
    got = {(cs.corner[i].tobytes(), np.float64(cs.edge[i]).tobytes()) for i in np.nonzero(cs.is_leaf)[0]}
    assert got == want


def test_count_scheme_grid_golden_and_manager_subset():
    from oracle import count_scheme_np as cnp

    g = load_golden("grid_L1_mixed.npz")
    coords, roots = cnp.grid_roots([g["points"]], 1)
    cs = cnp.count_scheme([g["points"]], coords, 1, 16, root_of=roots)
    assert _count_scheme_leaf_rows(cs, 0, len(g["points"])) == dict(golden_canon(g, "k16"))
    assert len(cs.edge) == g["k16_counts"][0]
    # L = 5, two poses, then a refinement driven by pose 1 only (the reference builds the new scheme
    # from scratch from the selected poses' counts, octree_manager.py:50-61)
    g = load_golden("grid_L5_two_poses.npz")
    poses = [g["points0"], g["points1"]]
    coords, roots = cnp.grid_roots(poses, 5)
    cs = cnp.count_scheme(poses, coords, 5, 24, root_of=roots)
    for p in range(2):
        assert _count_scheme_leaf_rows(cs, p, len(poses[p])) == dict(golden_canon(g, f"p{p}"))
    # (a voxel without points of pose 1 keeps an unsplit scheme there: same rule, count 0)
    cs = cnp.count_scheme(poses, coords, 5, 6, root_of=roots, scheme_poses=[1])
    og = onp.OGrid(5)
    for p in range(2):
        og.insert_points(p, poses[p])
    og.subdivide(6, [1])
    for p in range(2):
        assert _count_scheme_leaf_rows(cs, p, len(poses[p])) == dict(canon_from_list(og.leaf_table(p)))


def test_count_scheme_manager_multi_pose_vs_recursive_oracle():
    from oracle import count_scheme_np as cnp

    P, n, K = 8, 5000, 256
    poses = [np.random.default_rng(100 + p).random((n, 3)) for p in range(P)]
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(P):
        om.insert_points(p, poses[p])
    om.subdivide(K)
    cs = cnp.count_scheme(poses, np.zeros((1, 3)), 1.0, K, chunk=1500)
    for p in range(P):
        want = {((np.asarray(v.corner, dtype=np.float64) + 0.0).tobytes(), np.float64(v.edge).tobytes()):
                tuple(sorted(v.idx.tolist())) for v in om.octrees[p].leaves()}
        assert _count_scheme_leaf_rows(cs, p, n) == want
        assert len(cs.edge) == om.n_nodes(p)
    with pytest.raises(RecursionError):
        cnp.count_scheme([np.tile(np.array([[0.3, 0.3, 0.3]]), (5, 1))], np.zeros((1, 3)), 1.0, 2)


# ------------------------------------------------------------------------------------------------
# map_leaf_points with functions that replace the leaf's cloud (octree.py:114-123)
# ------------------------------------------------------------------------------------------------
def map_fn_bbox_corners(points):
    return np.vstack([points.min(axis=0), points.max(axis=0)])


def map_fn_halve_and_shift(points):
    return points * 0.5 + 10.0


def map_fn_triple(points):
    return np.vstack([points, points + 0.001, points.min(axis=0, keepdims=True)])


def canon_rows(table):
    """[(corner, edge, (n,3) rows)] -> [((corner bytes, edge bytes), rows sorted lexicographically as bytes)]"""
    out = []
    for corner, edge, rows in table:
        r = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 3)
        r = r[np.lexsort((r[:, 2], r[:, 1], r[:, 0]))]
        out.append((((np.asarray(corner, dtype=np.float64) + 0.0).tobytes(), np.float64(edge).tobytes()), r.tobytes()))
    return out


def golden_rows(g, prefix):
    off = np.concatenate(([0], np.cumsum(g[f"{prefix}_sizes"])))
    return canon_rows([(g[f"{prefix}_corners"][i], g[f"{prefix}_edges"][i], g[f"{prefix}_rows"][off[i]:off[i + 1]])
                       for i in range(len(g[f"{prefix}_edges"]))])


def run_map_transform_sequence(grid, snap):
    """the sequence tests/golden/make_golden.py:gen_grid_map_transform recorded from the reference"""
    grid.map_leaf_points(map_fn_bbox_corners, [0])
    snap("bbox")
    grid.map_leaf_points(map_fn_triple)
    snap("triple")
    grid.map_leaf_points(map_fn_halve_and_shift, [1])
    snap("shift")
    grid.filter([lambda pts: len(pts) > 6])
    snap("filtered")


def test_grid_map_leaf_points_transform():
    g = load_golden("grid_map_transform.npz")
    og = onp.OGrid(1)
    for p in range(2):
        og.insert_points(p, g[f"points{p}"])
    og.subdivide(int(g["K"]))

    def snap(tag):
        for p in range(2):
            assert canon_rows(og.leaf_rows(p)) == golden_rows(g, f"{tag}_p{p}")
            assert [og.n_nodes(p), og.n_leaves(p), og.n_points(p)] == list(g[f"{tag}_p{p}_counts"])

    run_map_transform_sequence(og, snap)


@pytest.mark.parametrize("name", ["h256", "h1024"])
def test_ransac_unique_maximiser_blocks_against_the_reference_plane(name):
    """Blocks on which ONE hypothesis attains the maximum (tests/golden/make_golden.py:gen_ransac_unique): the
    reference's tie race cannot choose, so its recorded best_plane / mask ARE the answer - the restatement must
    give the same f32 plane bits (north_star: |delta normal| <= 1e-5), count and mask on every block."""
    g = load_golden(f"ransac_unique_{name}.npz")
    cloud, sizes, hyp, thr = g["cloud"], g["block_sizes"], g["hypotheses"], float(g["threshold"])
    mask, count, plane, index, tied = rnp.evaluate(cloud, sizes, hyp, thr, details=True)
    assert all(len(t) == 1 for t in tied)
    assert np.array_equal(count, g["ref_max_inliers"])
    assert np.max(np.abs(plane[:, :3].astype(np.float64) - g["ref_plane"][:, :3])) <= 1e-5
    assert np.array_equal(plane.view(np.uint32), g["ref_plane"].view(np.uint32))
    assert np.array_equal(mask, g["mask"])
