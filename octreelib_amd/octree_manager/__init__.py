from octreelib_amd.octree_manager.octree_manager import OctreeManager

__all__ = ["OctreeManager"]
