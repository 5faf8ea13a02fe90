"""
Voxel value types (reference: internal/voxel.py:12-95).

Equality / hash are on (tuple(corner_min), edge_length) and equal voxels share one `.id`,
handed out by a process-global first-seen registry (voxel.py:19,29-32).  The registry is
kept, but ids are assigned on first ACCESS, so building 10^5 leaves does not pay 10^5 dict
insertions; only id equality is part of the contract (test_multi_pose.py:177-182), not the
numeric values.
"""

import itertools
from typing import Optional

import numpy as np

from octreelib_amd.internal.interfaces import WithID

__all__ = ["Voxel", "VoxelBase"]

_static_voxel_id_map = {}


def _voxel_id(corner_min, edge_length) -> int:
    key = (tuple(np.asarray(corner_min).tolist()), float(edge_length))
    got = _static_voxel_id_map.get(key)
    if got is None:
        got = len(_static_voxel_id_map)
        _static_voxel_id_map[key] = got
    return got


class VoxelBase(WithID):
    __slots__ = ("_corner_min", "_edge_length", "_lazy_id")

    def __init__(self, corner_min, edge_length):
        self._corner_min = corner_min
        self._edge_length = edge_length
        self._lazy_id = None

    @property
    def id(self):
        if self._lazy_id is None:
            self._lazy_id = _voxel_id(self._corner_min, self._edge_length)
        return self._lazy_id

    def __hash__(self):
        return hash((tuple(np.asarray(self._corner_min).tolist()), float(self._edge_length)))

    def __eq__(self, other):
        return bool(np.all(np.asarray(self.corner_min) == np.asarray(other.corner_min))) and (
            self.edge_length == other.edge_length
        )

    @property
    def corner_min(self):
        return self._corner_min

    @property
    def edge_length(self):
        return self._edge_length

    @property
    def corner_max(self):
        return self.corner_min + self.edge_length

    @property
    def all_corners(self):
        return [
            self._corner_min + offset
            for offset in itertools.product([0, self._edge_length], repeat=3)
        ]


class Voxel(VoxelBase):
    """Voxel with a point cloud."""

    __slots__ = ("_points",)

    def __init__(self, corner_min, edge_length, points: Optional[np.ndarray] = None):
        super().__init__(corner_min, edge_length)
        self._points = points if points is not None else np.empty((0, 3), dtype=float)

    def get_points(self):
        return np.array(self._points, dtype=float, copy=True).reshape(-1, 3)

    def insert_points(self, points):
        self._points = np.vstack([self.get_points(), np.asarray(points, dtype=float).reshape(-1, 3)])

    @property
    def n_points(self):
        return len(self._points)
