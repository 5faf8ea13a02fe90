"""
Lazy API objects over the flat device tables: leaf views (what get_leaf_points returns) and the
host-side helpers shared by Octree / OctreeManager / Grid for the callable-driven operations
(filter, map_leaf_points, apply_mask), which the reference defines on arbitrary Python
callables (octree/octree.py:102-142) and which therefore run on the host over device-produced
leaf arrays.
"""

from typing import Callable, Iterable, List, Sequence

import numpy as np

from octreelib_amd.internal.voxel import Voxel


class LeafView(Voxel):
    """One octree leaf: same duck type as the reference's cached OctreeNode / Voxel
    (corner_min, edge_length, id, n_points, get_points(), corner_max, all_corners).
    A snapshot: it keeps the coordinates array of the build it was created from."""

    __slots__ = ("_xyz", "_start", "_size", "node")

    def __init__(self, corner_min, edge_length, xyz, start, size, node):
        # (start, size, node: Python ints - the callers hand over .tolist() values; this constructor runs
        #  once per leaf, 4 * 10^5 times for a 10 M point grid)
        self._corner_min = corner_min
        self._edge_length = edge_length
        self._id = None  # resolved on first access (internal/voxel.py)
        self._points = None
        self._xyz = xyz
        self._start = start
        self._size = size
        self.node = node

    def get_points(self):
        if self._size == 0:
            return np.empty((0, 3), dtype=float)
        return self._xyz[self._start : self._start + self._size].copy()

    @property
    def n_points(self):
        return self._size

    def __repr__(self):
        return f"LeafView(corner_min={self._corner_min!r}, edge_length={self._edge_length!r}, n_points={self._size})"


def _node_corner(forest, node):
    nd = forest.nodes
    c = nd["corner"][node]
    if nd["depth"][node] == 0 and forest.mode == 0:
        # Grid managers are created with np.array(voxel_coordinates): an int64 corner
        # (grid/grid.py:100-106); children get float corners (octree.py:186)
        return forest.voxels[nd["voxel"][node]].copy()
    return c.copy()


def _node_corners(forest, nodes: np.ndarray):
    """_node_corner for many nodes at once: a list of per-leaf corner arrays (each its own copy)."""
    nd = forest.nodes
    corners = list(nd["corner"][nodes])          # rows of a fresh (m, 3) array
    if forest.mode == 0:
        roots = np.nonzero(nd["depth"][nodes] == 0)[0]
        if len(roots):
            vox = forest.voxels[nd["voxel"][nodes[roots]]]   # int64 corners of the root leaves
            for i, row in zip(roots.tolist(), vox):
                corners[i] = row
    return corners


def leaf_views(forest, slot: int, non_empty: bool = True) -> List[LeafView]:
    """Leaves of one pose in the reference's order (voxel lexicographic, cached-leaf order)."""
    nd = forest.nodes
    blk = forest.blocks
    if non_empty:
        ids = forest.slot_blocks(slot)
        if len(ids) == 0:
            return []
        xyz = forest.xyz
        nodes = blk["node"][ids]
        return [
            LeafView(c, e, xyz, s, z, n)
            for c, e, s, z, n in zip(_node_corners(forest, nodes), nd["edge"][nodes], blk["start"][ids].tolist(),
                                     blk["size"][ids].tolist(), nodes.tolist())
        ]
    # all leaves, empty ones included: host-side ordering from the node table
    order = all_leaves_order(forest, slot)
    xyz = forest.xyz
    ids = forest.slot_blocks(slot)
    # the pose's block of every listed leaf (none: an empty leaf) by one sorted search, not a dictionary per call
    starts = np.zeros(len(order), dtype=np.int64)
    sizes = np.zeros(len(order), dtype=np.int64)
    if len(ids) and len(order):
        bn = blk["node"][ids]
        by = np.argsort(bn, kind="stable")
        sbn = bn[by]
        pos = np.minimum(np.searchsorted(sbn, order), len(sbn) - 1)
        hit = sbn[pos] == order
        src = ids[by][pos]
        starts[hit] = blk["start"][src][hit]
        sizes[hit] = blk["size"][src][hit]
    starts, sizes = starts.tolist(), sizes.tolist()
    corners = _node_corners(forest, order)
    return [LeafView(c, e, xyz, s, z, n)
            for c, e, s, z, n in zip(corners, nd["edge"][order], starts, sizes, order.tolist())]


def preorder_rank(nd) -> np.ndarray:
    """DFS-preorder rank of every internal node among the internal nodes of its voxel
    (level-synchronous, vectorised): the order in which the reference splits nodes inside one
    subdivide / subdivide_as call (octree.py:20-53)."""
    n = len(nd["depth"])
    fc = nd["first_child"]
    depth = nd["depth"]
    internal = fc >= 0
    nint = np.zeros(n, dtype=np.int64)
    max_d = int(depth.max()) if n else 0
    for d in range(max_d, -1, -1):
        ids = np.nonzero(internal & (depth == d))[0]
        if len(ids) == 0:
            continue
        kids = fc[ids][:, None] + np.arange(8)[None, :]
        nint[ids] = 1 + nint[kids].sum(axis=1)
    rank = np.zeros(n, dtype=np.int64)
    for d in range(0, max_d + 1):
        ids = np.nonzero(internal & (depth == d))[0]
        if len(ids) == 0:
            continue
        kids = fc[ids][:, None] + np.arange(8)[None, :]
        sub = nint[kids]
        offs = np.cumsum(sub, axis=1) - sub
        rank[kids] = rank[ids][:, None] + 1 + offs
    return rank


def all_leaves_order(forest, slot: int) -> np.ndarray:
    """Node ids of ALL leaves (empty ones too) of the voxels a pose lives in, in the order of
    the reference's cached-leaf list (octree_base.py:152-158, octree.py:183-191)."""
    nd = forest.nodes
    ranks = forest.slot_voxel_ranks(slot) if forest.mode == 0 else np.array([0])
    leaves = np.nonzero(nd["first_child"] < 0)[0]
    leaves = leaves[np.isin(nd["voxel"][leaves], ranks)]
    if len(leaves) == 0:
        return leaves
    prank = preorder_rank(nd)
    par = nd["parent"][leaves]
    has_par = par >= 0
    p = np.where(has_par, par, 0)
    e0 = forest.slot_epoch[slot]
    eff = np.where(has_par, np.maximum(nd["epoch"][p], e0), 0)
    pr = np.where(has_par, prank[p], 0)
    digit = np.where(has_par, leaves - nd["first_child"][p], 0)
    key = np.lexsort((digit, pr, eff, nd["voxel"][leaves]))
    return leaves[key]


def positions_of_slot(forest, slot: int):
    """(starts, sizes) of the slot's non-empty leaves in cached-leaf order."""
    blk = forest.blocks
    ids = forest.slot_blocks(slot)
    return blk["start"][ids], blk["size"][ids]


def apply_mask_slot(forest, slot: int, mask):
    """Octree.apply_mask (octree.py:265-274): the mask runs over the pose's non-empty leaves in
    cached-leaf order."""
    mask = np.asarray(mask).astype(bool).reshape(-1)
    starts, sizes = positions_of_slot(forest, slot)
    keep = np.ones(forest.n_ord, dtype=np.uint8)
    off = 0
    for s, z in zip(starts.tolist(), sizes.tolist()):
        keep[s : s + z] = mask[off : off + z]
        off += z
    forest.apply_host_mask(keep)


def filter_slots(forest, slots: Iterable[int], criteria: Sequence[Callable]):
    """OctreeNode.filter (octree.py:102-112): a leaf whose points fail any criterion is emptied.
    Point-count criteria run on the device (no download of the cloud); arbitrary callables are the
    caller's Python code and are evaluated on the host over the leaf arrays."""
    from octreelib_amd.criteria import try_count_interval

    iv = try_count_interval(criteria)
    if iv is not None:
        forest.filter_count(list(slots), iv[0], iv[1])
        return
    xyz = forest.xyz
    keep = np.ones(forest.n_ord, dtype=np.uint8)
    changed = False
    for slot in slots:
        starts, sizes = positions_of_slot(forest, slot)
        for s, z in zip(starts.tolist(), sizes.tolist()):
            pts = xyz[s : s + z]
            if not all([c(pts) for c in criteria]):
                keep[s : s + z] = 0
                changed = True
    if changed:
        forest.apply_host_mask(keep)


def map_slots(forest, slots: Iterable[int], function: Callable):
    """OctreeNode.map_leaf_points (octree.py:114-123): every non-empty leaf of the selected poses keeps
    whatever the function returns for its cloud.  The function is the caller's Python code and runs on the
    host over the leaf arrays.  A result that is a SELECTION of the leaf's own rows (the RANSAC-like use)
    becomes a device compaction - the points keep their identity and insertion order; anything else (fewer
    rows, more rows, moved rows - rows may leave the leaf's cube, as upstream allows) replaces the contents
    of the leaves (octl_forest_set_contents)."""
    xyz = forest.xyz
    blk = forest.blocks
    keep = np.ones(forest.n_ord, dtype=np.uint8)
    results = {}          # block id -> the (m, 3) array the function returned
    selection = True
    for slot in slots:
        for b in forest.slot_blocks(slot).tolist():
            s, z = int(blk["start"][b]), int(blk["size"][b])
            pts = xyz[s : s + z]
            res = np.ascontiguousarray(np.asarray(function(pts.copy()), dtype=np.float64))
            if res.size % 3 != 0 or (res.ndim == 2 and res.shape[1] != 3 and res.size):
                raise ValueError(
                    f"map_leaf_points: the function must return an (n, 3) point cloud, got shape {res.shape}")
            res = res.reshape(-1, 3)
            results[b] = res
            if not selection:
                continue
            # rows matched against the leaf's own rows by value (+0.0 folds -0.0 into +0.0 on both sides;
            # equal rows are handed out in storage order), one dictionary per leaf
            where = {}
            for i in range(z - 1, -1, -1):
                where.setdefault((pts[i] + 0.0).tobytes(), []).append(i)
            sel = np.zeros(z, dtype=np.uint8)
            for row in res:
                hit = where.get((row + 0.0).tobytes())
                if not hit:
                    selection = False
                    break
                sel[hit.pop()] = 1
            keep[s : s + z] = sel
    if selection:
        forest.apply_host_mask(keep)
        return
    # contents replaced: every block in storage order, untouched ones with their own rows
    nodes, slots_out, sizes, rows = [], [], [], []
    for b in range(len(blk["node"])):
        r = results.get(b)
        if r is None:
            s, z = int(blk["start"][b]), int(blk["size"][b])
            r = xyz[s : s + z]
        if len(r) == 0:
            continue
        nodes.append(int(blk["node"][b]))
        slots_out.append(int(blk["slot"][b]))
        sizes.append(len(r))
        rows.append(r)
    forest.set_contents(np.asarray(nodes, dtype=np.int32), np.asarray(slots_out, dtype=np.int32),
                        np.asarray(sizes, dtype=np.int32),
                        np.vstack(rows) if rows else np.empty((0, 3), dtype=np.float64))
