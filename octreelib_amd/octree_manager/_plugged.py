"""
The reference's plug seam, served on the host (reference: grid/grid_base.py:66-87, grid/grid.py:100-106,
octree_manager/octree_manager.py:21-66,161-171).

The device path keeps a whole manager - every pose of one cube - in ONE forest and never instantiates an octree
object per pose.  A caller who configures his OWN `octree_type` expects exactly those objects to exist and to be
called; for him the manager falls back to what the reference does: one `octree_type(config, corner, edge)` per
pose plus a scheme octree that synchronises the subdivision, all driven through the octree's public interface
(insert_points / subdivide / subdivide_as / filter / map_leaf_points / get_points / get_leaf_points / counters /
apply_mask).  Slow, host-driven and correct; octreelib_amd's own Octree - and therefore any subclass that does not
override those methods - still builds on the device, one small forest per (voxel, pose).
"""

from typing import Callable, Dict, List, Optional

import numpy as np


class PluggedPoses:
    """{pose -> octree_type instance} + scheme octree of one manager (octree_manager.py:21-34)."""

    def __init__(self, octree_type, octree_config, corner_min, edge_length):
        self._make = lambda: octree_type(octree_config, np.asarray(corner_min), edge_length)
        self.octrees: Dict[int, object] = {}
        self.scheme = self._make()
        self._subdivided = False   # (an unsplit scheme has nothing to hand down: subdivide_as would be a no-op)

    # octree_manager.py:161-171: a pose that arrives after a subdivide inherits the scheme's shape
    def insert_points(self, pose_number: int, points):
        if pose_number not in self.octrees:
            self.octrees[pose_number] = self._make()
        tree = self.octrees[pose_number]
        tree.insert_points(points)
        if self._subdivided:
            tree.subdivide_as(self.scheme)

    # octree_manager.py:36-66: scheme from the union of the selected poses, then EVERY pose is forced to it
    def subdivide(self, criteria: List[Callable], pose_numbers: Optional[List[int]] = None):
        chosen = list(self.octrees) if pose_numbers is None else list(pose_numbers)
        self.scheme = self._make()
        clouds = [self.octrees[p].get_points() for p in chosen]   # (KeyError for an unknown pose, as upstream)
        if clouds:
            self.scheme.insert_points(np.vstack(clouds))
        self.scheme.subdivide(criteria)
        self.scheme.filter([lambda _points: False])                # keep the shape, drop the points
        self._subdivided = True
        for tree in self.octrees.values():
            tree.subdivide_as(self.scheme)

    def map_leaf_points(self, function: Callable, pose_numbers=None):
        for p in (list(self.octrees) if pose_numbers is None else pose_numbers):
            if p in self.octrees:
                self.octrees[p].map_leaf_points(function)

    def filter(self, criteria: List[Callable], pose_numbers=None):
        for p in (list(self.octrees) if pose_numbers is None else pose_numbers):
            self.octrees[p].filter(criteria)

    def get_leaf_points(self, non_empty=True, pose_number=None):
        if pose_number is None:
            out = []
            for tree in self.octrees.values():
                out.extend(tree.get_leaf_points(non_empty))
            return out
        return self.octrees[pose_number].get_leaf_points(non_empty) if pose_number in self.octrees else []

    def get_points(self, pose_number=None):
        if pose_number is None:
            parts = [t.get_points() for t in self.octrees.values()]
            return np.vstack(parts) if parts else np.empty((0, 3), dtype=float)
        return self.octrees[pose_number].get_points() if pose_number in self.octrees else np.empty((0, 3), dtype=float)

    @staticmethod
    def _counter(tree, name):
        v = getattr(tree, name)
        return int(v() if callable(v) else v)   # (properties on Octree, methods on some user types)

    def count(self, name, pose_number=None):
        if pose_number is None:
            return sum(self._counter(t, name) for t in self.octrees.values())
        return self._counter(self.octrees[pose_number], name) if pose_number in self.octrees else 0

    def apply_mask(self, mask, pose_number):
        if pose_number in self.octrees:
            self.octrees[pose_number].apply_mask(mask)
