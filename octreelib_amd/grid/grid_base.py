"""Config types and abstract surface of the grid (reference: grid/grid_base.py:17-211)."""

from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from enum import Enum
from typing import Generic, List, Type

import numpy as np

from octreelib_amd.internal.typing import T
from octreelib_amd.octree import Octree, OctreeBase, OctreeConfig, OctreeConfigBase
from octreelib_amd.octree_manager import OctreeManager

__all__ = ["GridVisualizationType", "VisualizationConfig", "GridConfigBase", "GridBase"]


class GridVisualizationType(Enum):
    POSE = "pose"
    VOXEL = "voxel"


@dataclass
class VisualizationConfig:
    """Kept for signature compatibility; Grid.visualize (k3d HTML export) is out of scope."""

    type: GridVisualizationType = GridVisualizationType.VOXEL
    point_size: float = 0.1
    line_width_size: float = 0.01
    line_color: int = 0xFF0000
    filepath: str = "visualization.html"
    seed: int = 0
    unused_voxels: List[int] = field(default_factory=list)


@dataclass
class GridConfigBase(ABC):
    octree_manager_type: Type[OctreeManager] = OctreeManager
    octree_type: Type[OctreeBase] = Octree
    octree_config: OctreeConfigBase = field(default_factory=OctreeConfig)
    debug: bool = False
    voxel_edge_length: float = 1
    corner: np.ndarray = field(default_factory=lambda: np.array(([0.0, 0.0, 0.0])))

    def __post_init__(self):
        # message texts are asserted verbatim by the reference's tests (test_grid.py:157-180)
        if not issubclass(self.octree_manager_type, OctreeManager):
            raise TypeError(
                f"Cannot use the provided octree manager type {self.octree_manager_type.__name__}. "
                "It has to be a subclass of octree_manager.OctreeManager."
            )
        if not issubclass(self.octree_type, OctreeBase):
            raise TypeError(
                f"Cannot use the provided octree type {self.octree_type.__name__}. "
                "It has to be a subclass of octree.OctreeBase."
            )


class GridBase(ABC, Generic[T]):
    def __init__(self, grid_config: GridConfigBase):
        self._grid_config = grid_config

    @abstractmethod
    def insert_points(self, pose_number, points): ...

    @abstractmethod
    def get_points(self, pose_number): ...

    @abstractmethod
    def subdivide(self, subdivision_criteria, pose_numbers=None): ...

    @abstractmethod
    def filter(self, filtering_criteria): ...

    @abstractmethod
    def map_leaf_points(self, function): ...

    @abstractmethod
    def map_leaf_points_cuda_ransac(self, poses_per_batch=1, threshold=0.01, hypotheses_number=1024): ...

    @abstractmethod
    def get_leaf_points(self, pose_number): ...

    @abstractmethod
    def visualize(self, config): ...

    @abstractmethod
    def n_nodes(self, pose_number): ...

    @abstractmethod
    def n_points(self, pose_number): ...

    @abstractmethod
    def n_leaves(self, pose_number): ...
