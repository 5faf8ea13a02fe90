"""
Single-pose octree over one cube (reference: octree/octree.py:14-295), device resident.

The tree is a flat scheme-node table + a leaf-ordered point permutation produced by HIP kernels
(octreelib_amd/csrc/build.hip); this class only maps the reference's methods onto it.
"""

from dataclasses import dataclass
from typing import Callable, Generic, List

import numpy as np

from octreelib_amd import _views
from octreelib_amd._engine import Forest
from octreelib_amd.criteria import try_count_threshold
from octreelib_amd.internal import T, Voxel
from octreelib_amd.octree.octree_base import OctreeBase, OctreeConfigBase, OctreeNodeBase

__all__ = ["OctreeNode", "Octree", "OctreeConfig"]


@dataclass
class OctreeConfig(OctreeConfigBase):
    pass


class Octree(OctreeBase, Generic[T]):
    """Octree(octree_config, corner_min, edge_length) - points of a single pose."""

    def __init__(self, octree_config: OctreeConfigBase, corner_min, edge_length):
        Voxel.__init__(self, corner_min, edge_length)
        self._config = octree_config
        self._forest = Forest(1, np.asarray(corner_min, dtype=np.float64), float(edge_length))
        self._slot = None

    # -- construction -------------------------------------------------------------------
    def insert_points(self, points):
        """octree.py:235-239.  Points added to an already subdivided tree descend the existing
        structure (octree.py:67-98) without triggering new splits."""
        if self._slot is None:
            self._slot = self._forest.add_pose(points)
        else:
            self._forest.extend_pose(self._slot, points)

    def subdivide(self, subdivision_criteria: List[Callable]):
        """octree.py:214-220 / 20-32.  A root that is already split holds no points itself, so
        a count criterion is false on it and the call changes nothing (upstream behaviour)."""
        k = try_count_threshold(subdivision_criteria)
        f = self._forest
        if self._slot is None:
            self._slot = f.add_pose(np.empty((0, 3)))
        if f.has_scheme and f.nodes["first_child"][0] >= 0:
            # upstream evaluates the criteria on the (empty) point array of the split root
            if not any([c(np.empty((0, 3), dtype=float)) for c in subdivision_criteria]):
                return
        if k is None:
            f.subdivide_callable(subdivision_criteria)
        else:
            f.subdivide(k)

    def subdivide_as(self, other_octree: "Octree"):
        """octree.py:222-227: split this octree wherever `other_octree` is split (the operation an
        OctreeManager applies to every pose).  The scheme is copied on the host, every point is
        placed on the device."""
        f = self._forest
        if self._slot is None:
            self._slot = f.add_pose(np.empty((0, 3)))
        f.adopt_scheme(other_octree._forest)

    # -- queries ------------------------------------------------------------------------
    def get_points(self):
        """octree.py:229-233: DFS order of the leaves = storage order."""
        if self._slot is None:
            return np.empty((0, 3), dtype=float)
        return self._forest.xyz.copy()

    def get_leaf_points(self, non_empty: bool = True) -> List[Voxel]:
        if self._slot is None:
            self._slot = self._forest.add_pose(np.empty((0, 3)))
        return _views.leaf_views(self._forest, self._slot, non_empty)

    @property
    def n_points(self):
        return 0 if self._slot is None else self._forest.n_points(self._slot)

    @property
    def n_leaves(self):
        return 0 if self._slot is None else self._forest.n_leaves(self._slot)

    @property
    def n_nodes(self):
        if self._slot is None:
            return 1
        self._forest.ensure_built()
        return 1 + 8 * int(self._forest.info.n_internal)

    # -- callable driven ----------------------------------------------------------------
    def filter(self, filtering_criteria: List[Callable]):
        if self._slot is not None:
            _views.filter_slots(self._forest, [self._slot], filtering_criteria)

    def map_leaf_points(self, function: Callable):
        if self._slot is not None:
            _views.map_slots(self._forest, [self._slot], function)

    def apply_mask(self, mask):
        if self._slot is not None:
            _views.apply_mask_slot(self._forest, self._slot, mask)


class OctreeNode(OctreeNodeBase):
    """Stand-alone node (reference: OctreeNode(corner_min, edge_length, octree_cached_leaves),
    octree_base.py:36-49).  It owns a device octree rooted at itself; the caller's
    `octree_cached_leaves` list is refreshed with the current leaves (empty ones included) after
    every structural change, like the list the reference nodes append themselves to."""

    def __init__(self, corner_min, edge_length, octree_cached_leaves: list):
        Voxel.__init__(self, corner_min, edge_length)
        self._tree = Octree(OctreeConfig(), corner_min, edge_length)
        self._cached_leaves = octree_cached_leaves
        self._cached_leaves.append(self)

    def _refresh_cache(self):
        self._cached_leaves[:] = self._tree.get_leaf_points(non_empty=False)

    def insert_points(self, points):
        self._tree.insert_points(points)

    def subdivide(self, subdivision_criteria):
        self._tree.subdivide(subdivision_criteria)
        self._refresh_cache()

    def subdivide_as(self, other):
        """octree.py:34-53: `other` is a node (or an Octree) over the same cube."""
        self._tree.subdivide_as(other._tree if isinstance(other, OctreeNode) else other)
        self._refresh_cache()

    def get_points(self):
        return self._tree.get_points()

    def get_leaf_points(self):
        """octree.py:125-135: fresh Voxel copies of the non-empty leaves, DFS order."""
        leaves = self._tree.get_leaf_points()
        leaves.sort(key=lambda v: v._start)
        return [Voxel(v.corner_min, v.edge_length, v.get_points()) for v in leaves]

    def filter(self, filtering_criteria):
        self._tree.filter(filtering_criteria)
        self._refresh_cache()

    def map_leaf_points(self, function):
        self._tree.map_leaf_points(function)
        self._refresh_cache()

    def apply_mask(self, mask):
        self._tree.apply_mask(mask)
        self._refresh_cache()

    @property
    def n_points(self):
        return self._tree.n_points

    @property
    def n_leaves(self):
        return self._tree.n_leaves

    @property
    def n_nodes(self):
        return self._tree.n_nodes
