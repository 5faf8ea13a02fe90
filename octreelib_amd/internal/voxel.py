"""
Voxel value types (reference: internal/voxel.py:12-95).

Two voxels are equal when corner and edge are equal, and equal voxels report the same `.id`
(voxel.py:19,29-41; the reference's tests compare ids of leaves with ids of freshly constructed
Voxels, test_multi_pose.py:177-182).  The reference registers every voxel in a process-global
dict at construction; here the registry is consulted on the first ACCESS of `.id`, so producing
10^5 leaf objects does not pay 10^5 dict insertions.  Only id equality is part of the contract,
not the numbers.
"""

import numpy as np

from octreelib_amd.internal.interfaces import WithID

__all__ = ["Voxel", "VoxelBase"]

# the 8 corner offsets in units of the edge, in itertools.product([0, e], repeat=3) order
_UNIT_CORNERS = np.array([(i >> 2 & 1, i >> 1 & 1, i & 1) for i in range(8)], dtype=np.int64)


class _Registry(dict):
    """(corner tuple, edge) -> dense id in first-seen order."""

    def id_of(self, key):
        try:
            return self[key]
        except KeyError:
            return self.setdefault(key, len(self))


_ids = _Registry()


class VoxelBase(WithID):
    __slots__ = ("_corner_min", "_edge_length")

    def __init__(self, corner_min, edge_length):
        self._corner_min, self._edge_length = corner_min, edge_length
        self._id = None  # resolved lazily, see the module docstring

    def _key(self):
        return tuple(np.asarray(self._corner_min).tolist()), float(self._edge_length)

    @property
    def id(self):
        if self._id is None:
            self._id = _ids.id_of(self._key())
        return self._id

    def __hash__(self):
        return hash(self._key())

    def __eq__(self, other):
        same_corner = np.asarray(self.corner_min) == np.asarray(other.corner_min)
        return bool(np.all(same_corner)) and self.edge_length == other.edge_length

    corner_min = property(lambda self: self._corner_min)
    edge_length = property(lambda self: self._edge_length)

    @property
    def corner_max(self):
        return self.corner_min + self.edge_length

    @property
    def all_corners(self):
        """The 8 corners; order of itertools.product([0, edge], repeat=3) (voxel.py:57-64)."""
        return [self._corner_min + unit * self._edge_length for unit in _UNIT_CORNERS]


class Voxel(VoxelBase):
    """A voxel holding an (N, 3) point array; get_points() hands out copies (voxel.py:85-89)."""

    __slots__ = ("_points",)

    def __init__(self, corner_min, edge_length, points=None):
        VoxelBase.__init__(self, corner_min, edge_length)
        self._points = np.empty((0, 3), dtype=float) if points is None else points

    def get_points(self):
        return np.array(self._points, dtype=float, copy=True).reshape(-1, 3)

    def insert_points(self, points):
        extra = np.asarray(points, dtype=float).reshape(-1, 3)
        self._points = np.concatenate((self.get_points(), extra), axis=0)

    n_points = property(lambda self: len(self._points))
