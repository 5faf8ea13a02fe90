"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (octreelib_amd/).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Count-only, level-synchronous restatement of the reference's count-driven subdivision, for inputs too
large for the node-by-node recursion of oracle/octree_np.py (BASELINE configs 3, 4, 5 at their stated
sizes: 10 M, 64 M, 125 M points).  Same arithmetic, same result, no point moves:

  * a node is split iff the scheme poses hold more than K points in it
    (OctreeNode.subdivide, octree/octree.py:20-32, with OctreeManager.subdivide building the scheme from
    the union of the selected poses, octree_manager/octree_manager.py:53-61, and forcing every pose to
    that shape, :65-66) - child counts never exceed the parent's, so the recursion's result does not
    depend on the order in which nodes are visited and all nodes of one depth can be decided together;
  * the child a point goes to is ``((p - corner_min) // (edge / 2)).astype(int)`` with the node's own
    corner and edge, child number ``4 ix + 2 iy + iz`` (octree.py:73-75,94-97);
  * children: edge ``edge / np.float64(2)``, corners ``corner + offset`` over
    ``itertools.product([0, child_edge], repeat=3)`` (octree.py:177-191);
  * top-level voxels of a Grid: ``((p - corner) // L * L).astype(int)`` (grid/grid.py:72-76), one
    independent cube per distinct voxel (grid.py:100-109,255-258).

Pinned: tests/test_oracle_golden.py checks this file against oracle/octree_np.py (itself pinned to the
reference's golden vectors) on single cubes, multi-pose managers with a pose subset, and grids.
"""

import itertools
from typing import List, Optional, Sequence

import numpy as np


class CountScheme:
    """Result: node table (roots first, then the 8 children of every split node, level by level) and,
    per pose, the leaf node of every point."""

    def __init__(self):
        self.corner = None       # (N, 3) f64
        self.edge = None         # (N,)   f64
        self.first_child = None  # (N,)   i64, -1 for leaves
        self.root = None         # (N,)   i64 root (top-level voxel) of every node
        self.leaf_of: List[np.ndarray] = []  # per pose: (n_p,) i64 node id of each point's leaf
        self.n_roots = 0

    @property
    def is_leaf(self):
        return self.first_child < 0

    def pose_counts(self, n_poses=None):
        """(N, P) points per node and pose (leaves only are non-zero)."""
        P = len(self.leaf_of) if n_poses is None else n_poses
        out = np.zeros((len(self.edge), P), dtype=np.int64)
        for p, lf in enumerate(self.leaf_of):
            out[:, p] = np.bincount(lf, minlength=len(self.edge))
        return out


def grid_roots(poses: Sequence[np.ndarray], L):
    """grid.py:72-81 for several poses at once: distinct top-level voxels in np.unique(axis=0) order
    (lexicographic) and, per pose, the root of every point."""

    def voxel_coords(pts):
        return ((np.asarray(pts, dtype=np.float64) - np.zeros(3)) // L * L).astype(int)  # grid.py:72-76

    lo = hi = None
    for pts in poses:  # (two passes so that only one int64 key per point is kept, not three)
        v = voxel_coords(pts)
        if len(v):
            lo = v.min(axis=0) if lo is None else np.minimum(lo, v.min(axis=0))
            hi = v.max(axis=0) if hi is None else np.maximum(hi, v.max(axis=0))
    span = hi - lo + 1
    # one int64 key per voxel, numeric order == lexicographic (x, y, z) order
    lins = []
    for pts in poses:
        v = voxel_coords(pts)
        lins.append(((v[:, 0] - lo[0]) * span[1] + (v[:, 1] - lo[1])) * span[2] + (v[:, 2] - lo[2]))
    uniq, inv = np.unique(np.concatenate(lins), return_inverse=True)
    z = uniq % span[2]
    y = (uniq // span[2]) % span[1]
    x = uniq // (span[2] * span[1])
    coords = np.stack([x + lo[0], y + lo[1], z + lo[2]], axis=1).astype(np.int64)
    roots, off = [], 0
    for l in lins:
        roots.append(inv[off : off + len(l)].astype(np.int64))
        off += len(l)
    return coords, roots


def count_scheme(poses: Sequence[np.ndarray], root_corner, root_edge, K: int,
                 root_of: Optional[Sequence[np.ndarray]] = None,
                 scheme_poses: Optional[Sequence[int]] = None, max_depth: int = 64,
                 chunk: int = 1 << 20) -> CountScheme:
    """poses[p]: (n_p, 3) f64.  root_corner: (V, 3) corners of the top-level cubes (V = 1 for a bare
    Octree / OctreeManager), root_edge: their edge.  root_of[p][i]: cube of point i (None: all 0).
    scheme_poses: the poses whose union drives the scheme (None: all)."""
    root_corner = np.asarray(root_corner, dtype=np.float64).reshape(-1, 3)
    V = len(root_corner)
    corner = [root_corner]
    edge = [np.full(V, np.float64(root_edge))]
    first_child = [np.full(V, -1, dtype=np.int64)]
    root = [np.arange(V, dtype=np.int64)]
    n_nodes = V
    in_scheme = set(range(len(poses))) if scheme_poses is None else set(scheme_poses)
    node_of = [np.zeros(len(p), dtype=np.int64) if root_of is None else np.asarray(root_of[i], dtype=np.int64).copy()
               for i, p in enumerate(poses)]
    level_lo, level_hi = 0, V  # nodes of the current depth
    for depth in range(max_depth + 1):
        cnt = np.zeros(level_hi - level_lo, dtype=np.int64)
        for p in in_scheme:
            nd = node_of[p]
            sel = nd >= level_lo
            cnt += np.bincount(nd[sel] - level_lo, minlength=level_hi - level_lo)
        split = np.nonzero(cnt > K)[0] + level_lo  # octree.py:26 with len(points) > K
        if len(split) == 0:
            break
        if depth == max_depth:
            raise RecursionError("maximum depth exceeded (duplicate points never separate)")
        cat_corner = np.concatenate(corner)
        cat_edge = np.concatenate(edge)
        # octree.py:177-191: 8 children per split node
        child_edge = cat_edge[split] / np.float64(2)
        new_corner = np.empty((len(split), 8, 3))
        for j, (ox, oy, oz) in enumerate(itertools.product([0, 1], repeat=3)):
            off = np.stack([ox * child_edge, oy * child_edge, oz * child_edge], axis=1)
            new_corner[:, j, :] = cat_corner[split] + off
        fc = np.concatenate(first_child)
        fc[split] = n_nodes + 8 * np.arange(len(split), dtype=np.int64)
        first_child = [fc]
        corner = [cat_corner, new_corner.reshape(-1, 3)]
        edge = [cat_edge, np.repeat(child_edge, 8)]
        first_child.append(np.full(8 * len(split), -1, dtype=np.int64))
        root = [np.concatenate(root), np.repeat(np.concatenate(root)[split], 8)]
        cat_corner = np.concatenate(corner)
        cat_edge = np.concatenate(edge)
        # octree.py:67-100: every pose's points in a split node descend one level
        for p in range(len(poses)):
            pts_all = np.asarray(poses[p], dtype=np.float64)
            nd_all = node_of[p]
            for s in range(0, len(nd_all), chunk):
                nd = nd_all[s : s + chunk]
                go = np.nonzero(fc[nd] >= 0)[0]
                if len(go) == 0:
                    continue
                par = nd[go]
                pts = pts_all[s : s + chunk][go]
                vi = ((pts - cat_corner[par]) // (cat_edge[par] / 2)[:, None]).astype(int)  # octree.py:73-75
                if vi.min() < 0 or vi.max() > 1:
                    raise ValueError("point outside the node's cube")
                nd[go] = fc[par] + 4 * vi[:, 0] + 2 * vi[:, 1] + vi[:, 2]  # octree.py:94-97
        level_lo, level_hi = n_nodes, n_nodes + 8 * len(split)
        n_nodes = level_hi
    out = CountScheme()
    out.corner = np.concatenate(corner)
    out.edge = np.concatenate(edge)
    out.first_child = np.concatenate(first_child)
    out.root = np.concatenate(root)
    out.leaf_of = node_of
    out.n_roots = V
    return out
