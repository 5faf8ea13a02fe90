"""
The reference's RANSAC operator, on an MI355X (reference: ransac/cuda_ransac.py:18-81; the
class keeps its upstream name so callers need not change).  evaluate() is one call into
liboctree_hip.so (octl_ransac_evaluate -> csrc/ransac.hip).
"""

import numpy as np

from octreelib_amd import _native as nat

__all__ = ["CudaRansac", "HipRansac"]

CUDA_THREADS = 1024


class CudaRansac:
    def __init__(self, threshold: float = 0.01, hypotheses_number: int = CUDA_THREADS,
                 initial_points_number: int = 6, ctx=None):
        self._threshold = float(threshold)
        self._threads_per_block = min(int(hypotheses_number), CUDA_THREADS)
        self._k = int(initial_points_number)
        # one table per operator object, drawn from NumPy's global generator (cuda_ransac.py:39-41)
        self._hypotheses = np.ascontiguousarray(np.random.random((self._threads_per_block, self._k)))
        self._ctx = ctx

    @property
    def random_hypotheses(self) -> np.ndarray:
        return self._hypotheses

    def evaluate(self, point_cloud, block_sizes, details: bool = False):
        """mask (M,) bool.  details=True also returns (planes (B,4) f32, best_count (B,) i32,
        best_index (B,) i32) - an extension; the reference only returns the mask."""
        ctx = self._ctx if self._ctx is not None else nat.get_context()
        cloud = nat.as_points(point_cloud)
        sizes = np.ascontiguousarray(np.asarray(block_sizes, dtype=np.int32).reshape(-1))
        M, B = len(cloud), len(sizes)
        mask = np.zeros(M, dtype=np.uint8)
        planes = np.zeros((B, 4), dtype=np.float32) if details else None
        counts = np.zeros(B, dtype=np.int32) if details else None
        index = np.zeros(B, dtype=np.int32) if details else None
        ctx.check(
            ctx.lib.octl_ransac_evaluate(
                ctx.handle, nat.ptr(cloud), M, nat.ptr(sizes), B, nat.ptr(self._hypotheses),
                self._threads_per_block, self._k, self._threshold, nat.ptr(mask),
                nat.ptr(planes), nat.ptr(counts), nat.ptr(index),
            )
        )
        mask = mask.astype(np.bool_)
        if details:
            return mask, planes, counts, index
        return mask


HipRansac = CudaRansac
