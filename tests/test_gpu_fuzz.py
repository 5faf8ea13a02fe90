"""
Randomised differential test: seeded sequences of Grid operations - inserts of several poses,
subdivide from a pose subset, late inserts that inherit the scheme, a refinement, RANSAC with
random batching - run on the HIP path and on the oracle; after EVERY operation the per-pose leaf
tables (corner bits, edge bits -> original index sets), the leaf LIST ORDER (history dependent,
octree_base.py:46-49) and the counters must agree exactly.

Sequences stay inside the parity domain of SURVEY.md 8a: integer edge length, distinct points,
re-subdivision only to an equal-or-finer scheme (all poses, K not larger than before), RANSAC last.
"""

import numpy as np
import pytest

from tests._util import assert_same_leaves, canon_from_list
from tests.test_gpu_parity import _oracle_grid_ransac, crit, index_map, views_table

pytestmark = pytest.mark.gpu


def _cloud(rng, n, edge, planar):
    """A pose: uniform background + tight clusters (+ planar patches for the RANSAC step)."""
    lo, hi = -2.0 * edge, 2.0 * edge
    parts = [rng.uniform(lo, hi, (n // 2, 3))]
    for _ in range(int(rng.integers(2, 6))):
        c = rng.uniform(lo * 0.9, hi * 0.9, 3)
        parts.append(c + rng.normal(0, edge * 10.0 ** rng.uniform(-4, -1), (n // 10, 3)))
    if planar:
        for _ in range(6):
            c = rng.uniform(lo * 0.8, hi * 0.8, 3)
            a, b = rng.uniform(-0.5, 0.5, 2)
            xy = rng.uniform(0, edge * 0.6, (n // 12, 2))
            z = a * xy[:, 0] + b * xy[:, 1] + rng.normal(0, 0.004, len(xy))
            parts.append(c + np.column_stack([xy, z]))
    pts = np.unique(np.clip(np.vstack(parts), lo, np.nextafter(hi, lo)), axis=0)
    rng.shuffle(pts)
    return pts


def _check(grid, og, poses, idx):
    for p in poses:
        got = canon_from_list(views_table(grid.get_leaf_points(p), idx[p]))
        assert_same_leaves(got, canon_from_list(og.leaf_table(p)))
        assert [grid.n_nodes(p), grid.n_leaves(p), grid.n_points(p)] == [og.n_nodes(p), og.n_leaves(p), og.n_points(p)]


@pytest.mark.parametrize("seed", range(16))
def test_random_operation_sequences_vs_oracle(seed):
    from octreelib_amd.grid import Grid, GridConfig
    from oracle import octree_np as onp

    rng = np.random.default_rng(1000 + seed)
    edge = int(rng.choice([1, 2, 5]))
    grid, og = Grid(GridConfig(voxel_edge_length=edge)), onp.OGrid(edge)
    poses, idx = {}, {}

    def insert(p):
        poses[p] = _cloud(rng, int(rng.integers(2000, 9000)), edge, planar=True)
        idx[p] = index_map(poses[p])
        grid.insert_points(p, poses[p])
        og.insert_points(p, poses[p])

    n0 = int(rng.integers(1, 4))
    for p in range(n0):
        insert(p)
    _check(grid, og, poses, idx)

    k1 = int(rng.integers(30, 250))
    subset = None
    if n0 > 1 and rng.random() < 0.7:
        subset = sorted(rng.choice(n0, int(rng.integers(1, n0 + 1)), replace=False).tolist())
    grid.subdivide(crit(k1), subset)
    og.subdivide(k1, subset)
    _check(grid, og, poses, idx)

    for p in range(n0, n0 + int(rng.integers(0, 3))):  # late poses inherit the scheme
        insert(p)
    _check(grid, og, poses, idx)

    k2 = int(rng.integers(8, k1 + 1))                   # equal or finer, driven by all poses
    grid.subdivide(crit(k2))
    og.subdivide(k2)
    _check(grid, og, poses, idx)

    if rng.random() < 0.5:
        insert(len(poses))
        _check(grid, og, poses, idx)

    H = int(rng.choice([64, 256, 1024]))
    ppb = int(rng.integers(1, 4))
    thr = float(rng.choice([0.01, 0.02]))
    np.random.seed(seed)
    table = np.random.random((H, 6))
    np.random.seed(seed)
    grid.map_leaf_points_cuda_ransac(poses_per_batch=ppb, threshold=thr, hypotheses_number=H,
                                     initial_points_number=6)
    _oracle_grid_ransac(og, poses, sorted(poses), table, thr, ppb)
    _check(grid, og, poses, idx)
