"""
Failure paths and long runs.

* Allocation failures: every device allocation of the library is a growth of a device buffer (devbuf_reserve), and
  octl_debug_fail_alloc(n) makes the n-th growth from now on fail exactly as a failed hipMalloc does.  For each step
  of a Grid's life - first build, a late pose, RANSAC, apply_mask, a second subdivide - n is swept over EVERYTHING the
  step allocates: the call must raise MemoryError with the library's message, and the same Grid, asked again, must
  arrive at the oracle's tables (the reference has no such path: NumPy raises MemoryError and leaves a half-built
  tree behind; the bar here is "the error is reported and nothing is silently corrupt").
* Soak: a few hundred scans of varying size through the Python classes with the pipelined feed - results stay
  those of the first pass over the same clouds, host and device memory stop growing once the pools are warm.
"""

import ctypes as C

import numpy as np
import pytest

from tests._util import assert_same_leaves, canon_from_list
from tests.test_gpu_parity import _oracle_grid_ransac, crit, index_map, views_table

pytestmark = pytest.mark.gpu


def _arm(nth):
    from octreelib_amd import _native as nat

    seen = C.c_int64(0)
    nat.get_context().check(nat.load().octl_debug_fail_alloc(int(nth), C.byref(seen)))
    return seen.value


def _cloud(seed, n, side=3):
    from octreelib_amd import synthetic

    return synthetic.planar_cloud(n, (side, side, side), seed=seed)


# (RANSAC last: it empties leaves, and a subdivide behind it can ask for a COARSER scheme than the poses have - the
#  reference's defective merge branch, outside the parity domain)
STEPS = ["insert+subdivide", "late pose", "second subdivide", "ransac+apply_mask"]


def _run_life(fail_step, nth, big=False):
    """One Grid through its whole life next to the oracle; the growth `nth` of step `fail_step` fails.
    Returns (raised, growths seen in that step).

    What the library promises after OCTL_E_NOMEM (include/octreelib_hip.h, octl_forest_build): the forest keeps
    its points and voxels and is left WITHOUT a scheme.  So the recovery of a failed step is "subdivide again":
    the tables must then be the oracle's; the leaf LIST ORDER, which depends on the history of the scheme
    (octree_base.py:46-49), is only compared on runs whose history was not cut."""
    from octreelib_amd import _native as nat
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    lib, ctx = nat.load(), nat.get_context()
    grid, og = Grid(GridConfig(voxel_edge_length=1)), onp.OGrid(1)
    # big: 27 voxels over four buckets, three levels deep
    poses = {0: _cloud(11, 9000), 1: _cloud(12, 5000)} if big else {0: _cloud(11, 2600, side=2), 1: _cloud(12, 1500, side=2)}
    idx = {p: index_map(c) for p, c in poses.items()}
    table_seed, H, thr = 5, 64, 0.01
    np.random.seed(table_seed)
    table = np.random.random((H, 6))
    state = {"ordered": True}

    def check(ps):
        for p in ps:
            got = canon_from_list(views_table(grid.get_leaf_points(p), idx[p]))
            assert_same_leaves(got, canon_from_list(og.leaf_table(p)), ordered=state["ordered"])
            assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == \
                   [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]

    def ransac():
        np.random.seed(table_seed)
        grid.map_leaf_points_cuda_ransac(poses_per_batch=2, threshold=thr, hypotheses_number=H)
        grid.n_points(0)

    def late_pose():
        grid.insert_points(1, poses[1])
        grid.n_leaves(1)

    # (device step, oracle step, recovery on the device after a failure, the same on the oracle, poses to compare)
    steps = [
        (lambda: (grid.subdivide(crit(24)), grid.n_leaves(0)), lambda: og.subdivide(24),
         lambda: grid.subdivide(crit(24)), lambda: None, [0]),
        (late_pose, lambda: og.insert_points(1, poses[1]),
         # the pose may or may not have been stored before the failure; the scheme is gone: rebuild it from pose 0
         lambda: ((1 in grid._slots) or grid.insert_points(1, poses[1]), grid.subdivide(crit(24), [0])),
         lambda: og.subdivide(24, [0]), [0, 1]),
        (lambda: (grid.subdivide(crit(10)), grid.n_leaves(0)), lambda: og.subdivide(10),
         lambda: grid.subdivide(crit(10)), lambda: None, [0, 1]),
        (ransac, lambda: _oracle_grid_ransac(og, poses, [0, 1], table, thr, 2), ransac, lambda: None, [0, 1]),
    ]

    grid.insert_points(0, poses[0])
    og.insert_points(0, poses[0])
    raised, seen = False, 0
    for s, (dev, ref, dev_recover, ref_recover, ps) in enumerate(steps):
        if s == fail_step:
            _arm(nth)
            try:
                dev()
            except MemoryError as e:
                raised = True
                assert "injected by octl_debug_fail_alloc" in str(e)
                assert b"injected" in lib.octl_last_error(ctx.handle)
            finally:
                seen = _arm(0)
            ref()
            if raised:
                # the grid still answers (points kept, no scheme or the old one) ...
                assert grid.n_points(0) in (len(poses[0]), og.n_points(0))
                dev_recover()       # ... and the same request, nothing injected, completes
                ref_recover()
                state["ordered"] = state["ordered"] and s in (0, 3)
        else:
            dev()
            ref()
        try:
            check(ps)           # ... in agreement with the oracle
        except AssertionError as e:
            raise AssertionError(f"after step {s} ({STEPS[s]}), failure injected in step {fail_step}: {e}") from e
    return raised, seen


@pytest.mark.parametrize("fail_step,big", [(0, False), (1, False), (2, False), (3, False), (0, True)],
                         ids=STEPS + ["insert+subdivide, four buckets"])
def test_allocation_failure_sweep(fail_step, big):
    nth, hits, problems = 1, 0, []
    while True:
        try:
            raised, seen = _run_life(fail_step, nth, big)
        except Exception as e:     # (the whole sweep is reported, not only its first casualty)
            import traceback

            where = [ln.strip() for ln in traceback.format_exc().splitlines() if "test_gpu_failures.py" in ln or "_util.py" in ln]
            problems.append(f"growth {nth}: {type(e).__name__}: {str(e)[:200]} @ {where[-3:]}")
            _arm(0)
            raised, seen = True, nth
            if len(problems) > 8:
                break
        if not raised:
            assert seen < nth          # the step finished: it allocates fewer than nth buffers
            break
        hits += 1
        nth += 1
        assert nth < 400, "the sweep does not terminate"
    assert not problems, "\n".join(problems)
    # a fresh Grid's first build allocates dozens of buffers; later steps at least their own scratch
    assert hits >= (8 if fail_step == 0 else 1), f"only {hits} allocation sites were visited"


def test_allocation_failure_in_the_c_abi_leaves_no_scheme_and_rebuilds():
    """The same at the C ABI: a failed octl_forest_build returns OCTL_E_NOMEM, the forest keeps its points and
    has no scheme (include/octreelib_hip.h), and the next build gives the tables of an undisturbed one."""
    from octreelib_amd import _native as nat

    lib, ctx = nat.load(), nat.get_context()
    pts = _cloud(3, 20000, side=4)

    def forest():
        fh = C.c_void_p()
        ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(np.zeros(3)), 1.0, C.byref(fh)))
        ctx.check(lib.octl_forest_add_pose(fh, nat.ptr(pts), len(pts), None))
        return fh

    def tables(fh):
        n = C.c_int64(0)
        ctx.check(lib.octl_forest_get_blocks(fh, 0, None, None, None, None, C.byref(n)))
        node, size = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        ctx.check(lib.octl_forest_get_blocks(fh, n.value, nat.ptr(node), None, None, nat.ptr(size), C.byref(n)))
        m = C.c_int64(0)
        ctx.check(lib.octl_forest_get_perm(fh, 0, None, C.byref(m)))
        perm = np.empty(m.value, np.int64)
        ctx.check(lib.octl_forest_get_perm(fh, m.value, nat.ptr(perm), C.byref(m)))
        return node.tobytes(), size.tobytes(), perm.tobytes()

    ref = forest()
    info = nat.BuildInfo()
    ctx.check(lib.octl_forest_build(ref, 32, None, 0, 0, 0, C.byref(info)))
    want = tables(ref)
    lib.octl_forest_destroy(ref)
    failures = 0
    for nth in range(1, 200):
        fh = forest()
        _arm(nth)
        rc = lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info))
        seen = _arm(0)
        if rc == nat.OCTL_OK:
            assert seen < nth
            assert tables(fh) == want
            lib.octl_forest_destroy(fh)
            break
        failures += 1
        assert rc == nat.OCTL_E_NOMEM
        assert b"injected" in lib.octl_last_error(ctx.handle)
        n = C.c_int64(-1)
        assert lib.octl_forest_get_blocks(fh, 0, None, None, None, None, C.byref(n)) == nat.OCTL_E_STATE
        ctx.check(lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info)))
        assert tables(fh) == want
        lib.octl_forest_destroy(fh)
    assert failures >= 8


def test_soak_scans_of_varying_size_no_drift_no_growth():
    """tools/soak.py in small: 240 scans of four sizes through the Python classes (a fresh Grid per scan, pipelined
    feed).  Every scan reproduces the count of its cloud's first pass; memory is flat after the warm-up."""
    import psutil

    import octreelib_amd as oa
    from octreelib_amd import MaxPoints, synthetic
    from octreelib_amd.grid import Grid, GridConfig

    n, scans = 400_000, 240
    clouds = [synthetic.planar_cloud(n - 37_000 * j, (12, 12, 12), seed=1, stream=j) for j in range(4)]
    stage = [oa.pinned_empty((n, 3)), oa.pinned_empty((n, 3))]
    proc = psutil.Process()
    hip = C.CDLL("libamdhip64.so")

    def dev_free():
        f, t = C.c_size_t(0), C.c_size_t(0)
        hip.hipMemGetInfo(C.byref(f), C.byref(t))
        return f.value

    def put(i):
        c = clouds[i % 4]
        stage[i & 1][: len(c)] = c
        return oa.upload_async(stage[i & 1][: len(c)])

    first, marks = {}, []
    nxt = put(0)
    for i in range(scans):
        cur = nxt
        grid = Grid(GridConfig(voxel_edge_length=1))
        grid.insert_points(0, cur)
        nxt = put(i + 1)
        grid.subdivide([MaxPoints(64)])
        np.random.seed(0)
        grid.map_leaf_points_cuda_ransac()
        got = (grid.n_points(0), grid.n_leaves(0) if i % 7 == 0 else None)
        want = first.setdefault(i % 4, got)
        assert got[0] == want[0] and (got[1] is None or want[1] is None or got[1] == want[1]), f"scan {i} drifted"
        grid._forest.close()
        cur.release()
        if (i + 1) % 40 == 0:
            marks.append((proc.memory_info().rss / 2 ** 20, dev_free() / 2 ** 20))
    nxt.wait()
    nxt.release()
    rss, free = [m[0] for m in marks], [m[1] for m in marks]
    assert rss[-1] - rss[1] < 48, f"host memory grows: {rss}"
    assert free[1] - free[-1] < 48, f"device memory grows: {free}"


def test_displaced_rows_survive_a_late_pose_and_refuse_a_re_placement():
    """map_leaf_points may move rows out of their leaf's cube; the reference keeps them in that leaf until the leaf
    is subdivided (octree.py:114-123, 94-98).  A pose inserted afterwards is placed incrementally (the displaced rows
    stay where they are: compared with the oracle); anything that would key every stored point by its coordinates
    again - here: more points for an existing pose - is refused instead of silently moving them."""
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp
    from tests.test_oracle_golden import canon_rows

    rng = np.random.default_rng(8)
    poses = [rng.random((800, 3)), rng.random((600, 3)), rng.random((300, 3))]
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    for p in range(2):
        m.insert_points(p, poses[p])
        om.insert_points(p, poses[p])
    m.subdivide(crit(40))
    om.subdivide(40)
    push = lambda pts: pts + np.array([0.3, 0.0, 0.0])      # most rows leave their (small) cubes
    m.map_leaf_points(push, [0])
    om.map_leaf_points(push, [0])
    keep = [lambda pts: len(pts) >= 3]
    m.filter(keep)
    om.filter(keep)
    m.insert_points(2, poses[2])                             # a late pose: inherits the scheme
    om.insert_points(2, poses[2])
    for p in range(3):
        got = canon_rows([(v.corner_min, v.edge_length, v.get_points()) for v in m.get_leaf_points(True, p)])
        want = canon_rows([(v.corner, v.edge, om.octrees[p].points[v.idx]) for v in om.octrees[p].leaves()])
        assert got == want
        assert [m.n_nodes(p), m.n_leaves(p), m.n_points(p)] == [om.n_nodes(p), om.n_leaves(p), om.n_points(p)]
    m.insert_points(1, rng.random((50, 3)))                  # extends pose 1: every stored point is placed again
    with pytest.raises((IndexError, ValueError), match="outside the cube of their leaf"):
        m.n_leaves(1)


def test_extend_pose_from_a_device_cloud_and_download_behind_an_upload():
    import octreelib_amd as oa
    from octreelib_amd import _native as nat
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from oracle import octree_np as onp
    from tests.test_oracle_golden import canon_rows

    rng = np.random.default_rng(9)
    a, b, c = rng.random((5000, 3)), rng.random((3000, 3)), rng.random((4000, 3))
    m = OctreeManager(Octree, OctreeConfig(), np.array([0.0, 0.0, 0.0]), 1.0)
    om = onp.OManager(np.array([0.0, 0.0, 0.0]), 1.0)
    stage = oa.pinned_empty((len(c), 3))
    stage[:] = c
    m.insert_points(0, oa.upload_async(a))
    m.insert_points(1, b)
    m.insert_points(0, oa.upload_async(stage))               # appended to pose 0, device to device
    for p, cl in ((0, a), (1, b), (0, c)):
        om.insert_points(p, cl)
    m.subdivide(crit(50))
    om.subdivide(50)
    for p in range(2):
        got = canon_rows([(v.corner_min, v.edge_length, v.get_points()) for v in m.get_leaf_points(True, p)])
        want = canon_rows([(v.corner, v.edge, om.octrees[p].points[v.idx]) for v in om.octrees[p].leaves()])
        assert got == want
    # octl_dev_download of a buffer whose upload is still in flight waits for it (on the device)
    ctx, lib = nat.get_context(), nat.load()
    big = oa.pinned_empty((2_000_000, 3))
    big[:] = rng.random((2_000_000, 3))
    cloud = oa.upload_async(big)
    back = np.empty_like(big)
    ctx.check(lib.octl_dev_download(ctx.handle, nat.ptr(back), cloud.ptr, back.nbytes))
    assert np.array_equal(back, big)
    cloud.release()


def test_set_contents_rejects_tables_that_are_not_the_forests_own():
    from octreelib_amd import _native as nat
    from octreelib_amd.grid import Grid, GridConfig

    pts = _cloud(4, 6000)
    grid = Grid(GridConfig(voxel_edge_length=1))
    grid.insert_points(0, pts)
    grid.subdivide(crit(30))
    f = grid._forest
    blk, xyz = f.blocks, f.xyz
    node, slot, size = blk["node"].copy(), blk["slot"].copy(), blk["size"].copy()
    rows = np.vstack([xyz[s : s + z] for s, z in zip(blk["start"].tolist(), size.tolist())])
    before = (grid.n_leaves(0), grid.n_points(0))
    internal = int(np.nonzero(f.nodes["first_child"] >= 0)[0][0])
    bad_node = node.copy()
    bad_node[0] = internal
    with pytest.raises(ValueError, match="not a leaf"):
        f.set_contents(bad_node, slot, size, rows)
    dup = node.copy()
    dup[1] = dup[0]
    with pytest.raises(ValueError, match="appears twice"):
        f.set_contents(dup, slot, size, rows)
    swapped = node.copy()
    swapped[[0, 1]] = swapped[[1, 0]]
    with pytest.raises(ValueError, match="out of the storage order"):
        f.set_contents(swapped, slot, size, rows)
    assert (grid.n_leaves(0), grid.n_points(0)) == before    # nothing was committed
    f.set_contents(node, slot, size, rows)                    # the forest's own table goes through
    assert (grid.n_leaves(0), grid.n_points(0)) == before


def test_scan_pipeline_on_two_contexts_gives_the_sequential_results():
    """octreelib_amd.ScanPipeline: 16 scans (four different clouds, sizes differ) taken alternately by two worker
    threads with a context each.  Every scan's surviving points and leaf count equal those of the plain sequential
    loop over the same clouds; errors inside a scan come back through its future."""
    import octreelib_amd as oa
    from octreelib_amd import MaxPoints, synthetic
    from octreelib_amd.grid import Grid, GridConfig

    clouds = [synthetic.planar_cloud(300_000 - 21_000 * j, (10, 10, 10), seed=1, stream=j) for j in range(4)]
    np.random.seed(3)
    table = np.random.random((512, 6))

    def fit(grid, i):
        grid.subdivide([MaxPoints(64)])
        grid.map_leaf_points_cuda_ransac(hypotheses=table)
        pts = grid.get_points(0)
        return grid.n_points(0), grid.n_leaves(0), float(pts.sum())

    want = []
    for c in clouds:
        g = Grid(GridConfig(voxel_edge_length=1))
        g.insert_points(0, c)
        want.append(fit(g, 0))
        g._forest.close()
    ring = [oa.pinned_empty((300_000, 3)) for _ in range(5)]   # (map keeps 3 scans in flight + the one being drawn)

    def scans():
        for i in range(16):
            c = clouds[i % 4]
            ring[i % 5][: len(c)] = c
            yield ring[i % 5][: len(c)]

    with oa.ScanPipeline(2) as pipe:
        got = list(pipe.map(scans(), fit))
        assert got == [want[i % 4] for i in range(16)]
        # a failing scan does not take the pipeline down
        bad = pipe.submit(np.full((10, 3), np.nan), lambda grid, i: grid.subdivide([MaxPoints(2)]))
        ok = pipe.submit(clouds[1], fit)
        with pytest.raises((ValueError, IndexError)):
            bad.result()
        assert ok.result() == want[1]
