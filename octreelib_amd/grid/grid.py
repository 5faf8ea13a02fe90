"""
Grid of top-level voxels, multi-pose (reference: grid/grid.py:21-362), device resident.

insert_points uploads a pose; the voxel bucketing, the synchronised count-driven subdivision,
the leaf ordering, the per-leaf RANSAC and the mask application all run as HIP kernels behind
the C ABI (include/octreelib_hip.h).
"""

import os
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import numpy as np

from octreelib_amd import _views
from octreelib_amd._engine import Forest
from octreelib_amd.criteria import try_count_threshold
from octreelib_amd.grid.grid_base import GridBase, GridConfigBase, VisualizationConfig
from octreelib_amd.internal.voxel import Voxel

__all__ = ["Grid", "GridConfig"]

RANSAC_MAX_HYPOTHESES = 1024  # the reference's CUDA_THREADS (ransac/cuda_ransac.py:15)
_CHECKS = os.environ.get("OCTREELIB_AMD_CHECKS", "0") not in ("", "0")   # invariants asserted (the GPU tests set it)


@dataclass
class GridConfig(GridConfigBase):
    pass


class Grid(GridBase):
    def __init__(self, grid_config: GridConfig):
        super().__init__(grid_config)
        L = grid_config.voxel_edge_length
        corner = np.asarray(grid_config.corner, dtype=np.float64).reshape(3)
        if np.any(corner != 0.0):
            raise NotImplementedError(
                "GridConfig.corner != 0 is outside the parity domain: the reference stores voxel "
                "corners relative to the grid corner but its octrees subtract them from absolute "
                "points (grid.py:96-105 vs octree.py:74) and fail on subdivide"
            )
        if float(L) <= 0 or float(L) != int(L):
            raise NotImplementedError(
                "voxel_edge_length must be a positive integer value: the reference truncates "
                "voxel coordinates with astype(int) (grid.py:72-76), merging fractional voxels"
            )
        # the reference's plug seam (grid_base.py:66-87, grid.py:100-106): it instantiates
        # octree_manager_type(octree_type, octree_config, corner, L) per top-level voxel.  With the package's own
        # types the whole grid is ONE device-resident forest.
        from octreelib_amd.octree import Octree
        from octreelib_amd.octree_manager import OctreeManager

        self._slots: Dict[int, int] = {}  # pose number -> slot
        self._plug = None
        self._forest = None
        if grid_config.octree_manager_type is not OctreeManager or grid_config.octree_type is not Octree:
            # the caller's own types: served on the host, one manager_type(octree_type, config, corner, L) per
            # top-level voxel as the reference instantiates them (grid/_plugged.py) - slow and correct
            from octreelib_amd.grid._plugged import PluggedGrid

            self._plug = PluggedGrid(grid_config)
        else:
            self._forest = Forest(0, corner, float(L))

    # grid.py:58-109
    def insert_points(self, pose_number: int, points):
        if self._plug is not None:
            return self._plug.insert_points(pose_number, points)
        if pose_number in self._slots:
            raise ValueError(f"Cannot insert points to existing pose {pose_number}")
        self._slots[pose_number] = self._forest.add_pose(points)

    # grid.py:244-258
    def subdivide(self, subdivision_criteria: List[Callable], pose_numbers: Optional[List[int]] = None):
        if self._plug is not None:
            return self._plug.subdivide(subdivision_criteria, pose_numbers)
        k = try_count_threshold(subdivision_criteria)
        scheme = None if pose_numbers is None else [self._slots[p] for p in pose_numbers]
        if k is None:
            self._forest.subdivide_callable(subdivision_criteria, scheme)
        else:
            self._forest.subdivide(k, scheme)

    # grid.py:217-232
    def get_leaf_points(self, pose_number: int, non_empty: bool = True) -> List[Voxel]:
        if self._plug is not None:
            return self._plug.get_leaf_points(pose_number, non_empty)
        return _views.leaf_views(self._forest, self._slots[pose_number], non_empty)

    # grid.py:234-242: all managers in first-creation order, DFS order inside a manager
    def get_points(self, pose_number: int):
        if self._plug is not None:
            return self._plug.get_points(pose_number)
        f = self._forest
        slot = self._slots[pose_number]
        blk = f.blocks
        sel = np.nonzero(blk["slot"] == slot)[0]
        if len(sel) == 0:
            return np.empty((0, 3), dtype=float)
        vox_rank = f.nodes["voxel"][blk["node"][sel]]
        creation = f.creation_ranks(f.voxels)[vox_rank]
        # managers in creation order, storage order inside one (= the DFS order of octree.get_points); a grid whose
        # voxels were created in voxel order - one pose, or poses over the same voxels - is in that order already.
        # INVARIANT the shortcut relies on: the block table is in storage order - `start` strictly ascending over the
        # non-empty blocks (forest.h: "block table of non-empty (leaf, pose) runs in storage order") - so that equal
        # creation ranks are already ordered by start.  OCTREELIB_AMD_CHECKS=1 verifies it.
        if _CHECKS and len(sel) > 1:
            assert np.all(np.diff(blk["start"][sel].astype(np.int64)) > 0), "block table out of storage order"
        if len(sel) > 1 and np.any(creation[1:] < creation[:-1]):
            sel = sel[np.lexsort((blk["start"][sel], creation))]
        return f.gather_blocks(sel)

    # grid.py:260-267
    def filter(self, filtering_criteria: List[Callable]):
        if self._plug is not None:
            return self._plug.filter(filtering_criteria)
        _views.filter_slots(self._forest, list(self._slots.values()), filtering_criteria)

    # grid.py:111-122
    def map_leaf_points(self, function: Callable, pose_numbers: Optional[List[int]] = None):
        if self._plug is not None:
            return self._plug.map_leaf_points(function, pose_numbers)
        if pose_numbers is None:
            slots = list(self._slots.values())
        else:
            slots = [self._slots[p] for p in pose_numbers if p in self._slots]
        _views.map_slots(self._forest, slots, function)

    # grid.py:124-215
    def map_leaf_points_cuda_ransac(
        self,
        poses_per_batch: int = 10,
        threshold: float = 0.01,
        hypotheses_number: int = 1024,
        initial_points_number: int = 6,
        *,
        hypotheses=None,
    ):
        """`hypotheses` (extension, keyword only): the (H, k) table itself instead of one drawn from NumPy's global
        generator - for callers that fit scans from several threads (octreelib_amd.ScanPipeline) and want every
        scan to see the same table without serialising on the generator."""
        if hypotheses is not None:
            hypotheses = np.ascontiguousarray(hypotheses, dtype=np.float64)
            if hypotheses.ndim != 2:
                raise ValueError("hypotheses must be an (H, k) table")
            hypotheses_number, initial_points_number = hypotheses.shape
        if threshold <= 0:
            raise ValueError("Threshold must be positive")
        if hypotheses_number < 1:
            raise ValueError("Number of RANSAC hypotheses must be positive")
        if hypotheses_number > RANSAC_MAX_HYPOTHESES:
            raise ValueError(
                "Number of RANSAC hypotheses must be <= 1024 "
                "because of the CUDA thread limit."
            )
        if self._plug is not None:
            return self._plug.ransac(poses_per_batch, threshold, min(hypotheses_number, RANSAC_MAX_HYPOTHESES),
                                     initial_points_number, hypotheses)
        f = self._forest
        n_poses = len(self._slots)
        if n_poses == 0:
            return
        # the hypothesis table: ONE draw from NumPy's global generator, shared by all leaves
        # and batches (ransac/cuda_ransac.py:39-41)
        if hypotheses is not None:
            table = hypotheses
        else:
            table = np.random.random((min(hypotheses_number, RANSAC_MAX_HYPOTHESES), initial_points_number))
        # batches are ranges of pose INDICES used as pose numbers (grid.py:149-157)
        for p in range(n_poses):
            if p not in self._slots:
                raise KeyError(p)
        if all(self._slots[p] == p for p in range(n_poses)):
            f.ransac_all(poses_per_batch, table, threshold)  # order + kernel on the device
        else:
            for i in range(0, n_poses, poses_per_batch):
                batch = range(i, min(i + poses_per_batch, n_poses))
                order = np.concatenate([f.slot_blocks(self._slots[p]) for p in batch])
                f.ransac_blocks(order, table, threshold)
        f.apply_device_mask()  # grid.py:203-215 -> apply_mask: outliers leave the tree

    def visualize(self, config: VisualizationConfig = VisualizationConfig()) -> None:
        raise NotImplementedError("Grid.visualize (k3d HTML export) is out of scope of this build")

    # grid.py:343-362
    def n_leaves(self, pose_number: int) -> int:
        if self._plug is not None:
            return self._plug.count("n_leaves", pose_number)
        return self._forest.n_leaves(self._slots[pose_number])

    def n_points(self, pose_number: int) -> int:
        if self._plug is not None:
            return self._plug.count("n_points", pose_number)
        return self._forest.n_points(self._slots[pose_number])

    def n_nodes(self, pose_number: int) -> int:
        if self._plug is not None:
            return self._plug.count("n_nodes", pose_number)
        return self._forest.n_nodes(self._slots[pose_number])
