"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (octreelib_amd/).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

NumPy restatement of the reference's insert / subdivide / multi-pose algorithm
(same algorithm class: Python recursion over nodes, NumPy floor_divide / unique /
argsort / split per internal node).  Each function cites the reference lines it
restates (paths relative to /root/reference).

Pinned against the reference itself: tests/golden/*.npz are produced by importing the
reference (tests/golden/make_golden.py) and test_oracle_golden.py checks this file
against every one of them, plus the hand-written known answers of the reference's own
tests (test/octree/*.py, test/grid/test_grid.py).

Deliberate, documented deviations from the reference's *behaviour*:
  * within-leaf point order: the reference regroups with ``argsort()`` (unstable
    quicksort, NumPy-build dependent — octree/octree.py:87, grid/grid.py:88).  Here the
    argsort is ``kind="stable"`` so the order inside a leaf is the original insertion
    order.  Only per-leaf *sets* are comparable with the reference.
  * the cached-leaf list (octree_base.py:152-158, octree.py:183) is an insertion-ordered
    dict instead of a Python list: identical order semantics, O(1) removal instead of the
    reference's O(#leaves) ``list.remove``.
  * coarsening in subdivide_as (octree.py:48-53) is buggy upstream; here a
    node whose scheme counterpart is a leaf is merged properly.
Points are tracked as *indices* into the pose's cloud, so results can be compared as
index sets.
"""

import itertools
from typing import Callable, Dict, List, Optional, Sequence, Union

import numpy as np

Criterion = Union[int, Sequence[Callable[[np.ndarray], bool]]]


class ONode:
    """One octree node.  Restates OctreeNode (octree/octree.py:19-201)."""

    __slots__ = ("corner", "edge", "idx", "children", "tree")

    def __init__(self, corner, edge, tree):
        self.corner = corner
        self.edge = edge
        self.idx = np.empty(0, dtype=np.int64)  # indices into tree.points
        self.children: Optional[List["ONode"]] = None
        self.tree = tree
        tree.cached[id(self)] = self  # octree_base.py:49 (append to the cached leaves)

    # octree.py:177-191
    def _generate_children(self):
        child_edge = self.edge / np.float64(2)
        offsets = itertools.product([0, child_edge], repeat=3)
        del self.tree.cached[id(self)]  # octree.py:183
        return [ONode(self.corner + off, child_edge, self.tree) for off in offsets]

    # octree.py:67-100
    def insert(self, idx: np.ndarray):
        if self.children is not None:
            pts = self.tree.points[idx]
            voxel_indices = ((pts - self.corner) // (self.edge / 2)).astype(int)
            if len(idx) and (voxel_indices.min() < 0 or voxel_indices.max() > 1):
                # the reference indexes self._children[child_id] with a wrong id here
                # (IndexError or a silently wrong child) — outside the parity domain.
                raise ValueError("point outside the node's cube")
            uniq, inv = np.unique(voxel_indices, axis=0, return_inverse=True)
            inv = inv.reshape(-1)
            order = inv.argsort(kind="stable")
            groups = np.split(idx[order], np.cumsum(np.bincount(inv))[:-1])
            for u, g in zip(uniq, groups):
                child_id = sum(2**i * e for i, e in enumerate(u[::-1]))  # octree.py:94-97
                self.children[child_id].insert(g)
        else:
            self.idx = np.concatenate([self.idx, idx])  # octree.py:100

    def _split(self):
        self.children = self._generate_children()
        mine, self.idx = self.idx, np.empty(0, dtype=np.int64)
        self.insert(mine)

    # octree.py:20-32
    def subdivide(self, crit: Criterion):
        if _criterion_true(crit, self.tree.points, self.idx):
            self._split()
            for c in self.children:
                c.subdivide(crit)

    # octree.py:34-53
    def subdivide_as(self, other: "ONode"):
        if other.children is not None and self.children is None:
            self._split()
        if other.children is not None:
            for a, b in zip(self.children, other.children):
                a.subdivide_as(b)
        elif self.children is not None:
            # proper merge (the reference's merge branch is defective)
            self.idx = self.get_idx()
            for c in self.children:
                c._remove_from_cache()
            self.children = None
            self.tree.cached[id(self)] = self

    def _remove_from_cache(self):
        if self.children is not None:
            for c in self.children:
                c._remove_from_cache()
        else:
            del self.tree.cached[id(self)]

    # octree.py:55-65 (DFS child order)
    def get_idx(self) -> np.ndarray:
        if self.children is None:
            return self.idx.copy()
        parts = [c.get_idx() for c in self.children]
        return np.concatenate(parts) if parts else np.empty(0, dtype=np.int64)

    # octree.py:155-164
    def n_nodes(self) -> int:
        if self.children is None:
            return 1
        return 1 + sum(c.n_nodes() for c in self.children)


def _criterion_true(crit: Criterion, points: np.ndarray, idx: np.ndarray) -> bool:
    """octree.py:26 — any(criterion(points)); an int K means ``len(points) > K``."""
    if isinstance(crit, (int, np.integer)):
        return len(idx) > crit
    pts = points[idx]
    return any([c(pts) for c in crit])


class OTree:
    """Single-pose octree.  Restates Octree (octree.py:203-295, octree_base.py:133-158)."""

    def __init__(self, corner, edge, points: Optional[np.ndarray] = None):
        self.corner = corner
        self.edge = edge
        self.points = (
            np.empty((0, 3), dtype=float) if points is None else np.asarray(points, dtype=float)
        )
        self.cached: Dict[int, ONode] = {}
        self.root = ONode(corner, edge, self)

    def insert_points(self, points: np.ndarray):
        points = np.asarray(points, dtype=float).reshape(-1, 3)
        start = len(self.points)
        self.points = np.vstack([self.points, points])
        self.root.insert(np.arange(start, start + len(points), dtype=np.int64))

    def subdivide(self, crit: Criterion):
        self.root.subdivide(crit)

    def subdivide_as(self, other: "OTree"):
        self.root.subdivide_as(other.root)

    def get_idx(self) -> np.ndarray:
        return self.root.get_idx()

    def get_points(self) -> np.ndarray:
        return self.points[self.get_idx()]

    # octree.py:256-263
    def leaves(self, non_empty: bool = True) -> List[ONode]:
        if non_empty:
            return [v for v in self.cached.values() if len(v.idx) != 0]
        return list(self.cached.values())

    # octree.py:265-274
    def apply_mask(self, mask: np.ndarray):
        start = 0
        for leaf in self.leaves(True):
            n = len(leaf.idx)
            leaf.idx = leaf.idx[mask[start : start + n]]
            start += n

    # octree.py:102-112, 222-228: a leaf whose points fail any criterion is emptied (all(...) over the
    # criteria; the criterion sees the leaf's point array)
    def filter(self, criteria):
        for leaf in self.cached.values():
            if not all([c(self.points[leaf.idx]) for c in criteria]):
                leaf.idx = np.empty(0, dtype=np.int64)

    # octree.py:114-123, 230-233: a non-empty leaf keeps whatever the function returns for its cloud (any number
    # of rows, anywhere).  The new rows are appended to the pose's point array and the leaf refers to them.
    def map_leaf_points(self, function):
        for leaf in self.cached.values():
            if len(leaf.idx):
                new = np.asarray(function(self.points[leaf.idx].copy()), dtype=float).reshape(-1, 3)
                start = len(self.points)
                self.points = np.vstack([self.points, new])
                leaf.idx = np.arange(start, start + len(new), dtype=np.int64)

    @property
    def n_points(self):
        return sum(len(v.idx) for v in self.cached.values())

    @property
    def n_leaves(self):
        return len(self.leaves(True))

    @property
    def n_nodes(self):
        return self.root.n_nodes()


class OManager:
    """Restates OctreeManager (octree_manager/octree_manager.py:12-180)."""

    def __init__(self, corner, edge):
        self.corner = corner
        self.edge = edge
        self.octrees: Dict[int, OTree] = {}
        self.scheme = OTree(corner, edge)  # :34

    # :161-171
    def insert_points(self, pose: int, points: np.ndarray):
        if pose not in self.octrees:
            self.octrees[pose] = OTree(self.corner, self.edge)
        self.octrees[pose].insert_points(points)
        self.octrees[pose].subdivide_as(self.scheme)

    # :36-66
    def subdivide(self, crit: Criterion, pose_numbers=None):
        if pose_numbers is None:
            pose_numbers = list(self.octrees.keys())
        self.scheme = OTree(self.corner, self.edge)
        union = [
            self.octrees[p].get_points() for p in pose_numbers if p in self.octrees
        ]  # reference raises KeyError for a pose absent from this voxel; treated as empty
        if union:
            self.scheme.insert_points(np.vstack(union))
        self.scheme.subdivide(crit)
        for v in self.scheme.cached.values():  # :63 filter([lambda _: False])
            v.idx = np.empty(0, dtype=np.int64)
        for p in self.octrees:
            self.octrees[p].subdivide_as(self.scheme)

    # octree_manager.py:68-83
    def map_leaf_points(self, function, pose_numbers=None):
        if pose_numbers is None:
            pose_numbers = list(self.octrees.keys())
        for p in pose_numbers:
            if p in self.octrees:
                self.octrees[p].map_leaf_points(function)

    # octree_manager.py:85-99
    def filter(self, criteria, pose_numbers=None):
        if pose_numbers is None:
            pose_numbers = list(self.octrees.keys())
        for p in pose_numbers:
            if p in self.octrees:
                self.octrees[p].filter(criteria)

    def n_nodes(self, pose):
        return self.octrees[pose].n_nodes if pose in self.octrees else 0

    def n_leaves(self, pose):
        return self.octrees[pose].n_leaves if pose in self.octrees else 0

    def n_points(self, pose):
        return self.octrees[pose].n_points if pose in self.octrees else 0


class OGrid:
    """Restates Grid (grid/grid.py:39-362) for the build-and-query path."""

    def __init__(self, voxel_edge_length=1, corner=None):
        self.L = voxel_edge_length
        self.corner = np.array([0.0, 0.0, 0.0]) if corner is None else corner
        self.pose_voxels: Dict[int, List[tuple]] = {}
        self.managers: Dict[tuple, OManager] = {}
        self.pose_points: Dict[int, np.ndarray] = {}
        # per (voxel, pose): map from the manager-local point index to the pose index
        self.local_to_pose: Dict[tuple, np.ndarray] = {}

    # grid.py:58-109
    def insert_points(self, pose: int, points: np.ndarray):
        if pose in self.pose_voxels:
            raise ValueError(f"Cannot insert points to existing pose {pose}")
        points = np.asarray(points, dtype=float).reshape(-1, 3)
        self.pose_voxels[pose] = []
        self.pose_points[pose] = points
        voxel_indices = ((points - self.corner) // self.L * self.L).astype(int)
        uniq, inv = np.unique(voxel_indices, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        order = inv.argsort(kind="stable")
        groups = np.split(order, np.cumsum(np.bincount(inv))[:-1])
        for coords, g in zip(uniq, groups):
            key = tuple(int(c) for c in coords)
            if key not in self.managers:
                self.managers[key] = OManager(np.array(coords), self.L)
            self.pose_voxels[pose].append(key)
            self.local_to_pose[(key, pose)] = g.astype(np.int64)
            self.managers[key].insert_points(pose, points[g])

    # grid.py:244-258
    def subdivide(self, crit: Criterion, pose_numbers=None):
        for m in self.managers.values():
            m.subdivide(crit, pose_numbers)

    # grid.py:217-232 — list of (corner f64[3], edge f64, sorted pose-point indices)
    def leaf_table(self, pose: int, non_empty: bool = True):
        out = []
        for key in self.pose_voxels[pose]:
            tree = self.managers[key].octrees[pose]
            l2p = self.local_to_pose[(key, pose)]
            for leaf in tree.leaves(non_empty):
                out.append(
                    (
                        np.asarray(leaf.corner, dtype=np.float64),
                        np.float64(leaf.edge),
                        l2p[leaf.idx],
                    )
                )
        return out

    def n_nodes(self, pose):
        return sum(m.n_nodes(pose) for m in self.managers.values())

    def n_leaves(self, pose):
        return sum(m.n_leaves(pose) for m in self.managers.values())

    def n_points(self, pose):
        return sum(m.n_points(pose) for m in self.managers.values())

    # grid.py:111-122
    def map_leaf_points(self, function, pose_numbers=None):
        for m in self.managers.values():
            m.map_leaf_points(function, pose_numbers)

    # leaves of a pose as (corner, edge, (n, 3) rows) - for clouds that map_leaf_points has replaced
    def leaf_rows(self, pose: int, non_empty: bool = True):
        out = []
        for key in self.pose_voxels[pose]:
            tree = self.managers[key].octrees[pose]
            for leaf in tree.leaves(non_empty):
                out.append((np.asarray(leaf.corner, dtype=np.float64), np.float64(leaf.edge), tree.points[leaf.idx]))
        return out

    # grid.py:260-267: every voxel's manager, all poses
    def filter(self, criteria):
        for m in self.managers.values():
            m.filter(criteria)

    # grid.py:203-215: mask is consumed per top voxel (per-pose voxel order), then per
    # non-empty cached leaf
    def apply_mask(self, pose: int, mask: np.ndarray):
        start = 0
        for key in self.pose_voxels[pose]:
            tree = self.managers[key].octrees[pose]
            n = tree.n_points
            tree.apply_mask(mask[start : start + n])
            start += n


def tree_leaf_table(tree: OTree, non_empty: bool = True):
    return [
        (np.asarray(v.corner, dtype=np.float64), np.float64(v.edge), v.idx.copy())
        for v in tree.leaves(non_empty)
    ]
