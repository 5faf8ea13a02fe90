"""
octreelib_amd - MI355X-native implementation of octreelib's point-cloud -> octree-grid
build-and-query path (Grid / OctreeManager / Octree / ransac), as a drop-in for that path:

    from octreelib_amd.grid import Grid, GridConfig
    from octreelib_amd.octree import Octree, OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager
    from octreelib_amd.ransac import CudaRansac

Python host code over a C ABI (include/octreelib_hip.h, ctypes) over hand-written HIP kernels
for gfx950.  There is no CPU fallback: without liboctree_hip.so and a GPU every operation
raises.
"""

from octreelib_amd.criteria import MaxPoints
from octreelib_amd.feed import DeviceCloud, ScanPipeline, pinned_empty, upload_async

__version__ = "0.1.0"
__all__ = ["MaxPoints", "DeviceCloud", "ScanPipeline", "pinned_empty", "upload_async", "__version__"]
