"""RANSAC operator (HIP)."""

from octreelib_amd.ransac.cuda_ransac import CudaRansac, HipRansac

__all__ = ["CudaRansac", "HipRansac"]
