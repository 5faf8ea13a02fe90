"""
Grid over the caller's own manager type (the reference's plug seam, grid/grid_base.py:66-87 and grid/grid.py:100-109):
`octree_manager_type(octree_type, octree_config, corner, L)` is instantiated once per top-level voxel and driven
through its public interface, as the reference does.  Only the voxel bucketing of a pose - the part that does not
involve the plug types - runs on the device (a throw-away forest without a scheme); everything else is the host
loop over the managers: slow, and exactly what a user of that seam asked for.
"""

from typing import Callable, Dict, List, Optional

import numpy as np

from octreelib_amd._engine import Forest
from octreelib_amd import _native as nat


class PluggedGrid:
    def __init__(self, config):
        self._cfg = config
        self._L = config.voxel_edge_length
        self._managers: Dict[tuple, object] = {}          # voxel coordinates -> manager, in creation order
        self._pose_voxels: Dict[int, List[tuple]] = {}    # pose -> its voxels in lexicographic order

    # grid.py:58-109
    def insert_points(self, pose_number: int, points):
        if pose_number in self._pose_voxels:
            raise ValueError(f"Cannot insert points to existing pose {pose_number}")
        pts = nat.as_points(points)
        self._pose_voxels[pose_number] = []
        if len(pts) == 0:
            return
        # voxel bucketing on the device: roots in lexicographic voxel order (np.unique(axis=0), grid.py:79-81),
        # a voxel's points in insertion order
        f = Forest(0, np.zeros(3), float(self._L))
        try:
            f.add_pose(pts)
            f.build(-1)
            voxels, blk, xyz = f.voxels.copy(), {k: v.copy() for k, v in f.blocks.items()}, f.xyz.copy()
            root_of_block = f.nodes["voxel"][blk["node"]]
        finally:
            f.close()
        for b in np.argsort(root_of_block, kind="stable"):
            coords = voxels[root_of_block[b]]   # (the reference's voxel "coordinates" q * L: what the table holds)
            key = tuple(int(c) for c in coords)
            if key not in self._managers:
                self._managers[key] = self._cfg.octree_manager_type(
                    self._cfg.octree_type, self._cfg.octree_config, np.array(coords), self._L)
            self._pose_voxels[pose_number].append(key)
            s, n = int(blk["start"][b]), int(blk["size"][b])
            self._managers[key].insert_points(pose_number, xyz[s : s + n])

    # grid.py:244-267,111-122
    def subdivide(self, criteria: List[Callable], pose_numbers: Optional[List[int]] = None):
        for m in self._managers.values():
            m.subdivide(criteria, pose_numbers)

    def filter(self, criteria: List[Callable]):
        for m in self._managers.values():
            m.filter(criteria)

    def map_leaf_points(self, function: Callable, pose_numbers: Optional[List[int]] = None):
        for m in self._managers.values():
            m.map_leaf_points(function, pose_numbers)

    # grid.py:217-242
    def get_leaf_points(self, pose_number: int, non_empty: bool = True):
        out = []
        for key in self._pose_voxels[pose_number]:
            out.extend(self._managers[key].get_leaf_points(non_empty, pose_number))
        return out

    def get_points(self, pose_number: int):
        parts = [m.get_points(pose_number) for m in self._managers.values()]
        parts = [p for p in parts if len(p)]
        return np.vstack(parts) if parts else np.empty((0, 3), dtype=float)

    # grid.py:124-215 with the device operator
    def ransac(self, poses_per_batch, threshold, hypotheses_number, initial_points_number, table=None):
        from octreelib_amd.ransac import CudaRansac

        n_poses = len(self._pose_voxels)
        if n_poses == 0:
            return
        # ONE operator, hence one table from NumPy's global generator, for all batches (grid.py:160-164)
        op = CudaRansac(threshold, hypotheses_number, initial_points_number)
        if table is not None:
            op._hypotheses = np.ascontiguousarray(table, dtype=np.float64)
        for i in range(0, n_poses, poses_per_batch):
            batch = list(range(i, min(i + poses_per_batch, n_poses)))
            clouds, sizes = [], []
            for p in batch:
                for leaf in self.get_leaf_points(p):
                    pts = leaf.get_points()
                    clouds.append(pts)
                    sizes.append(len(pts))
            if not clouds:
                continue
            mask = op.evaluate(np.vstack(clouds), np.array(sizes, dtype=np.int32))
            at = 0
            for p in batch:
                for key in self._pose_voxels[p]:
                    m = self._managers[key]
                    n = int(m.n_points(p))
                    m.apply_mask(mask[at : at + n], p)
                    at += n

    # grid.py:343-362
    def count(self, name: str, pose_number: int) -> int:
        return sum(int(getattr(m, name)(pose_number)) for m in self._managers.values())
