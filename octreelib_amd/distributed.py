"""
Multi-GPU sharding of a Grid by top-level voxel (one process per GPU).

The reference is single process / single device; what makes the path shard is that every
top-level voxel is an independent OctreeManager (grid/grid.py:56,100-109) and neither
Grid.subdivide (grid.py:255-258) nor the per-leaf RANSAC has a cross-voxel dependency.  Every
rank keeps the voxels `voxel_owner(q) == rank`; points are routed to their owners with one
all-to-all (RCCL grouped send/recv over xGMI inside liboctree_hip.so: csrc/route.hip), after
which insert / subdivide / RANSAC are purely local.  Counters are sums over ranks.

`voxel_owner_np` is the host mirror of the device hash (used by the CPU tests of the
partition logic and by callers that want to pre-partition on the host).
"""

import ctypes as C
from typing import Callable, Optional

import numpy as np

from octreelib_amd import _native as nat
from octreelib_amd._engine import Forest

__all__ = ["voxel_owner_np", "voxel_indices_np", "owned_voxel_ids", "ShardedGrid"]

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def voxel_indices_np(points: np.ndarray, L: float) -> np.ndarray:
    """Integer top-level voxel index floor(p / L) per axis (grid/grid.py:72-76 with corner 0)."""
    return np.floor_divide(np.asarray(points, dtype=np.float64), float(L)).astype(np.int64)


def voxel_owner_np(q: np.ndarray, n_ranks: int) -> np.ndarray:
    """Owner rank of voxel indices q (n,3) int64 - bit-identical to octl_voxel_owner /
    voxel_owner_hash (csrc/ref_arith.h)."""
    q = np.asarray(q, dtype=np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        h = q[:, 0] * np.uint64(0x9E3779B97F4A7C15)
        h = h ^ (q[:, 1] * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(0x165667B19E3779F9) + (h << np.uint64(6)) + (h >> np.uint64(2)))
        h = h ^ (q[:, 2] * np.uint64(0xD6E8FEB86659FD93) + np.uint64(0x27D4EB2F165667C5) + (h << np.uint64(6)) + (h >> np.uint64(2)))
        h = h ^ (h >> np.uint64(30))
        h = h * np.uint64(0xBF58476D1CE4E5B9)
        h = h ^ (h >> np.uint64(27))
        h = h * np.uint64(0x94D049BB133111EB)
        h = h ^ (h >> np.uint64(31))
    return (h % np.uint64(max(n_ranks, 1))).astype(np.int32)


def owned_voxel_ids(dims, rank: int, n_ranks: int) -> np.ndarray:
    """Linear ids (x slowest) of the voxels of a dims[0] x dims[1] x dims[2] scene anchored at the
    origin that `rank` of `n_ranks` owns."""
    d = np.asarray(dims, dtype=np.int64)
    lin = np.arange(int(d.prod()), dtype=np.int64)
    q = np.stack([lin // (d[1] * d[2]), (lin // d[2]) % d[1], lin % d[2]], axis=1)
    return lin[voxel_owner_np(q, n_ranks) == rank]


class ShardedGrid:
    """One rank's shard of a grid of 1-pose-at-a-time clouds.

    comm_broadcast: callable(bytes or None, src=0) -> bytes, used once to distribute the RCCL
    unique id (e.g. a torch.distributed / MPI broadcast - plumbing, not part of the data path).
    """

    def __init__(self, voxel_edge_length: float, rank: int, n_ranks: int,
                 comm_broadcast: Optional[Callable] = None, device: Optional[int] = None):
        self.ctx = nat.Context(nat.default_device() if device is None else device)
        self.lib = self.ctx.lib
        self.rank, self.n_ranks = int(rank), int(n_ranks)
        self.L = float(voxel_edge_length)
        self._has_comm = self.n_ranks > 1 or comm_broadcast is not None
        if self._has_comm:
            if comm_broadcast is None:
                raise ValueError("comm_broadcast is required for n_ranks > 1")
            uid = None
            if self.rank == 0:
                buf = (C.c_uint8 * nat.UNIQUE_ID_BYTES)()
                self.ctx.check(self.lib.octl_comm_unique_id(C.cast(buf, C.c_void_p)))
                uid = bytes(buf)
            uid = comm_broadcast(uid)
            idbuf = (C.c_uint8 * nat.UNIQUE_ID_BYTES).from_buffer_copy(uid)
            self.ctx.check(self.lib.octl_comm_init(self.ctx.handle, self.n_ranks, self.rank,
                                                   C.cast(idbuf, C.c_void_p)))
        self.forest = Forest(0, np.zeros(3), self.L, ctx=self.ctx)
        self._corner = np.zeros(3)

    def upload_async(self, points):
        """Start the upload of this rank's part of a pose on THIS grid's context (octreelib_amd.upload_async with
        the right context): hand the result to insert_points while the previous pose is still being built."""
        from octreelib_amd.feed import DeviceCloud

        return DeviceCloud(points, ctx=self.ctx)

    def insert_points(self, points, index_base: int = 0) -> int:
        """Route this rank's part of a pose to the owners and insert what this rank receives.
        Returns the number of points received.  Global point index = index_base + local index.
        `points`: a host array, or a DeviceCloud from self.upload_async (its copy may still be in flight: the
        routing kernels are ordered behind it on the device).  The staging buffers of host arrays come from the
        context's pool and are given back one pose later - no allocation, no free and no host wait per pose."""
        from octreelib_amd.feed import DeviceCloud

        if isinstance(points, DeviceCloud):
            if points.ctx is not self.ctx:
                raise ValueError("the DeviceCloud lives on another context: use ShardedGrid.upload_async")
            cloud, own = points, False
        else:
            cloud, own = DeviceCloud(points, ctx=self.ctx), True
        n_recv = C.c_int64(0)
        self.ctx.check(self.lib.octl_route_points(self.ctx.handle, cloud.ptr, None, cloud.n, int(index_base),
                                                  nat.ptr(self._corner), self.L, C.byref(n_recv), None))
        # (octl_route_points returns after its stream has drained: the cloud has been consumed)
        if own:
            cloud.release()
        self.forest.add_pose_routed(n_recv.value)
        return n_recv.value

    def routed_global_indices(self) -> np.ndarray:
        n = C.c_int64(0)
        self.ctx.check(self.lib.octl_route_get_gidx(self.ctx.handle, 0, None, C.byref(n)))
        g = np.empty(n.value, dtype=np.int64)
        self.ctx.check(self.lib.octl_route_get_gidx(self.ctx.handle, n.value, nat.ptr(g), C.byref(n)))
        return g

    def subdivide(self, K: int, scheme_slots=None):
        self.forest.subdivide(K, scheme_slots)

    def ransac(self, table: np.ndarray, threshold: float, poses_per_batch: int = 10):
        self.forest.ransac_all(poses_per_batch, table, threshold)
        self.forest.apply_device_mask()

    def global_counters(self, slot: int):
        """(n_nodes, n_leaves, n_points) summed over all ranks (grid.py:343-362)."""
        v = np.array([self.forest.n_nodes(slot), self.forest.n_leaves(slot), self.forest.n_points(slot)],
                     dtype=np.int64)
        self.ctx.check(self.lib.octl_comm_allreduce_i64(self.ctx.handle, nat.ptr(v), 3))
        return tuple(int(x) for x in v)

    def close(self):
        self.forest.close()
        if self._has_comm:
            self.lib.octl_comm_destroy(self.ctx.handle)
        self.ctx.close()
