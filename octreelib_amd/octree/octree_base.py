"""Abstract surface of the single-pose octree (reference: octree/octree_base.py:13-242)."""

from abc import ABC, abstractmethod
from dataclasses import dataclass

from octreelib_amd.internal.voxel import Voxel

__all__ = ["OctreeConfigBase", "OctreeBase", "OctreeNodeBase"]


@dataclass
class OctreeConfigBase(ABC):
    """debug is kept for signature compatibility; the reference never reads it."""

    debug: bool = True


class _TreeOps(ABC):
    """Operations shared by nodes and trees."""

    @property
    @abstractmethod
    def n_nodes(self): ...

    @property
    @abstractmethod
    def n_leaves(self): ...

    @property
    @abstractmethod
    def n_points(self): ...

    @abstractmethod
    def filter(self, filtering_criteria): ...

    @abstractmethod
    def map_leaf_points(self, function): ...

    @abstractmethod
    def subdivide(self, subdivision_criteria): ...

    @abstractmethod
    def subdivide_as(self, other): ...

    @abstractmethod
    def get_points(self): ...

    @abstractmethod
    def apply_mask(self, mask): ...


class OctreeNodeBase(Voxel, _TreeOps):
    def __init__(self, corner_min, edge_length, octree_cached_leaves):
        """octree_base.py:36-49: a node registers itself in its octree's list of leaves."""
        Voxel.__init__(self, corner_min, edge_length)
        self._cached_leaves = octree_cached_leaves
        self._cached_leaves.append(self)

    @abstractmethod
    def get_leaf_points(self): ...


class OctreeBase(Voxel, _TreeOps):
    def __init__(self, octree_config, corner_min, edge_length):
        """octree_base.py:136-150."""
        Voxel.__init__(self, corner_min, edge_length)
        self._config = octree_config

    @abstractmethod
    def get_leaf_points(self, non_empty: bool): ...

    @abstractmethod
    def insert_points(self, points): ...
