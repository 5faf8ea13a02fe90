"""
One cube, many poses, one shared subdivision scheme (reference: octree_manager/
octree_manager.py:12-180).  The scheme is built on the device from the union of the selected
poses' points and every pose is placed in it (octreelib_amd/csrc/build.hip).
"""

from typing import Callable, Dict, List, Optional, Type

import numpy as np

from octreelib_amd import _views
from octreelib_amd._engine import Forest
from octreelib_amd.criteria import try_count_threshold
from octreelib_amd.internal.voxel import Voxel, VoxelBase
from octreelib_amd.octree.octree_base import OctreeBase, OctreeConfigBase

__all__ = ["OctreeManager"]


class OctreeManager(VoxelBase):
    def __init__(
        self,
        octree_type: Type[OctreeBase],
        octree_config: OctreeConfigBase,
        corner_min,
        edge_length: float,
    ):
        super().__init__(corner_min, edge_length)
        self._octree_type = octree_type
        self._octree_config = octree_config
        self._slots: Dict[int, int] = {}  # pose number -> slot (insertion order)
        # the plug seam (grid_base.py:66-68): a caller's OWN octree type is instantiated per pose and driven through
        # its public interface on the host (octree_manager/_plugged.py); octreelib_amd's Octree means "one forest"
        from octreelib_amd.octree import Octree

        self._plug = None
        self._forest = None
        if octree_type is not Octree:
            from octreelib_amd.octree_manager._plugged import PluggedPoses

            self._plug = PluggedPoses(octree_type, octree_config, corner_min, edge_length)
        else:
            self._forest = Forest(1, np.asarray(corner_min, dtype=np.float64), float(edge_length))

    # octree_manager.py:161-171
    def insert_points(self, pose_number: int, points):
        if self._plug is not None:
            return self._plug.insert_points(pose_number, points)
        if pose_number not in self._slots:
            self._slots[pose_number] = self._forest.add_pose(points)
        else:
            self._forest.extend_pose(self._slots[pose_number], points)

    # octree_manager.py:36-66
    def subdivide(self, subdivision_criteria: List[Callable], pose_numbers: Optional[List[int]] = None):
        if self._plug is not None:
            return self._plug.subdivide(subdivision_criteria, pose_numbers)
        k = try_count_threshold(subdivision_criteria)
        if pose_numbers is None:
            scheme = None
        else:
            scheme = [self._slots[p] for p in pose_numbers]  # KeyError for an unknown pose, as upstream
        if k is None:
            self._forest.subdivide_callable(subdivision_criteria, scheme)
        else:
            self._forest.subdivide(k, scheme)

    def _selected(self, pose_numbers):
        if pose_numbers is None:
            return list(self._slots.values())
        return [self._slots[p] for p in pose_numbers if p in self._slots]

    def map_leaf_points(self, function: Callable, pose_numbers: Optional[List[int]] = None):
        if self._plug is not None:
            return self._plug.map_leaf_points(function, pose_numbers)
        _views.map_slots(self._forest, self._selected(pose_numbers), function)

    def filter(self, filtering_criteria: List[Callable], pose_numbers: Optional[List[int]] = None):
        if self._plug is not None:
            return self._plug.filter(filtering_criteria, pose_numbers)
        slots = list(self._slots.values()) if pose_numbers is None else [self._slots[p] for p in pose_numbers]
        _views.filter_slots(self._forest, slots, filtering_criteria)

    # octree_manager.py:101-119 (note the argument order)
    def get_leaf_points(self, non_empty: bool = True, pose_number: Optional[int] = None) -> List[Voxel]:
        if self._plug is not None:
            return self._plug.get_leaf_points(non_empty, pose_number)
        if pose_number is None:
            return sum((_views.leaf_views(self._forest, s, non_empty) for s in self._slots.values()), [])
        if pose_number in self._slots:
            return _views.leaf_views(self._forest, self._slots[pose_number], non_empty)
        return []

    def get_points(self, pose_number: Optional[int] = None):
        if self._plug is not None:
            return self._plug.get_points(pose_number)
        f = self._forest
        if pose_number is None:
            parts = [self.get_points(p) for p in self._slots]
            return np.vstack(parts) if parts else np.empty((0, 3), dtype=float)
        if pose_number not in self._slots:
            return np.empty((0, 3), dtype=float)
        # octree.get_points(): DFS order of the leaves = storage order
        blk = f.blocks
        sel = np.nonzero(blk["slot"] == self._slots[pose_number])[0]
        return f.gather_blocks(sel)

    def n_points(self, pose_number: Optional[int] = None) -> int:
        if self._plug is not None:
            return self._plug.count("n_points", pose_number)
        if pose_number is None:
            return sum(self._forest.n_points(s) for s in self._slots.values())
        if pose_number in self._slots:
            return self._forest.n_points(self._slots[pose_number])
        return 0

    def n_leaves(self, pose_number: int) -> int:
        if self._plug is not None:
            return self._plug.count("n_leaves", pose_number)
        if pose_number in self._slots:
            return self._forest.n_leaves(self._slots[pose_number])
        return 0

    def n_nodes(self, pose_number: int) -> int:
        if self._plug is not None:
            return self._plug.count("n_nodes", pose_number)
        if pose_number in self._slots:
            self._forest.ensure_built()
            return 1 + 8 * int(self._forest.info.n_internal)
        return 0

    # octree_manager.py:173-180
    def apply_mask(self, mask, pose_number: int):
        if self._plug is not None:
            return self._plug.apply_mask(mask, pose_number)
        if pose_number in self._slots:
            _views.apply_mask_slot(self._forest, self._slots[pose_number], mask)
