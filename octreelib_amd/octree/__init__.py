"""Octree, OctreeNode, OctreeConfig and their abstract bases."""

from octreelib_amd.octree.octree_base import OctreeBase, OctreeConfigBase, OctreeNodeBase
from octreelib_amd.octree.octree import Octree, OctreeConfig, OctreeNode

__all__ = ["OctreeConfigBase", "OctreeBase", "OctreeNodeBase", "OctreeNode", "Octree", "OctreeConfig"]
