"""Internal value types of the drop-in API (Voxel, VoxelBase, WithID, Point, PointCloud, T)."""

from octreelib_amd.internal.interfaces import WithID
from octreelib_amd.internal.point import Point, PointCloud
from octreelib_amd.internal.typing import T
from octreelib_amd.internal.voxel import Voxel, VoxelBase

__all__ = ["T", "Voxel", "VoxelBase", "Point", "PointCloud", "WithID"]
