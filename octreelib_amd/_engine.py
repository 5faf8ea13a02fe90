"""
Host-side view of one device-resident forest (include/octreelib_hip.h `octl_forest`): the flat
tables behind Grid / OctreeManager / Octree.  Everything heavy (bucketing, subdivision, leaf
ordering, RANSAC, compaction) happens in HIP kernels; this module only keeps NumPy mirrors of
the small result tables and turns them into the reference's API objects lazily.
"""

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

from octreelib_amd import _native as nat


class Forest:
    """mode 0 = Grid (top-level voxels of edge `edge`), mode 1 = one cube (Octree / Manager)."""

    def __init__(self, mode: int, corner, edge, ctx: Optional[nat.Context] = None):
        self.ctx = ctx if ctx is not None else nat.get_context()
        self.lib = self.ctx.lib
        self.mode = mode
        corner = np.ascontiguousarray(np.asarray(corner, dtype=np.float64).reshape(3))
        h = C.c_void_p()
        self.ctx.check(
            self.lib.octl_forest_create(self.ctx.handle, mode, nat.ptr(corner), float(edge), C.byref(h))
        )
        self.handle = h
        self.n_slots = 0
        self.slot_sizes: List[int] = []      # points ever inserted per slot
        self.slot_epoch: List[int] = []      # build epoch at which the slot's octrees were created
        self.epoch = 0                       # number of subdivide() calls so far
        self.has_scheme = False              # a K-driven scheme exists
        self._dirty = True                   # points were added since the last build
        self.info = None
        self._n_ord_pending = False
        self.n_ord = 0
        self._invalidate()
        # grid bookkeeping that must survive rebuilds: voxels each pose was inserted into and
        # the order in which voxels were first created (Grid.__octrees dict order, grid.py:56)
        self.slot_voxel_keys: List[Optional[np.ndarray]] = []
        self._edge_int = int(edge) if mode == 0 else 1
        self._cube = (tuple(np.asarray(corner, dtype=np.float64).tolist()), float(edge))
        self._creation_codes = np.empty(0, dtype=np.int64)   # packed voxel keys, creation order
        self._code_origin = None                              # voxel index the host-side codes are relative to
        self._member_next = 0                                 # first slot whose voxels have not been captured yet
        self._member_pending = []                             # (slot, voxel ids at capture time): keys not formed yet
        self._device_clouds = []                             # DeviceCloud objects whose buffers the store may read
        self._in_place = None                                # the DeviceCloud the store reads in place, if any

    # -- lifetime ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.octl_forest_destroy(self.handle)
            self.handle = C.c_void_p()
            self._device_clouds = []
            self._in_place = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def reads_in_place(self, cloud) -> bool:
        return self._in_place is cloud

    def _invalidate(self):
        self._nodes = None
        self._voxels = None
        self._blocks = None
        self._order = None
        self._xyz = None
        self._perm = None
        self._slot_blocks = None
        self._counts = None
        self._internal = None

    # -- points -----------------------------------------------------------------------------
    def add_pose(self, points) -> int:
        from octreelib_amd.feed import DeviceCloud

        if isinstance(points, DeviceCloud):
            # a cloud that is (being) uploaded already: read in place when it is the first pose, copied on
            # the device otherwise; the forest keeps the object alive while it reads its buffer
            import weakref

            first = sum(self.slot_sizes) == 0 and points.n > 0   # (the library adopts iff its store is empty)
            self._device_clouds.append(points)
            slot = self.add_pose_device(points.ptr, points.n, adopt=True)
            if first:   # (only the first pose of an empty forest is read in place; later ones are copied)
                self._in_place = points
                points._readers.append(weakref.ref(self))
            else:
                self._in_place = None   # (the store grew: it is the forest's own now)
            return slot
        pts = nat.as_points(points)
        slot = C.c_int32(-1)
        self.ctx.check(self.lib.octl_forest_add_pose(self.handle, nat.ptr(pts), len(pts), C.byref(slot)))
        self._in_place = None
        self._register_slot(len(pts))
        return slot.value

    def add_pose_device(self, dptr, n: int, adopt: bool = False) -> int:
        """A cloud that is in device memory already.  adopt=True: an empty forest reads the caller's buffer in
        place (no copy; the buffer must stay alive and unchanged until the forest is cleared or closed)."""
        slot = C.c_int32(-1)
        fn = self.lib.octl_forest_add_pose_adopt if adopt else self.lib.octl_forest_add_pose_device
        self.ctx.check(fn(self.handle, dptr, int(n), C.byref(slot)))
        self._register_slot(int(n))
        return slot.value

    def add_pose_routed(self, n: int) -> int:
        slot = C.c_int32(-1)
        self.ctx.check(self.lib.octl_forest_add_pose_routed(self.handle, C.byref(slot)))
        self._register_slot(int(n))
        return slot.value

    def _register_slot(self, n: int):
        self.n_slots += 1
        self.slot_sizes.append(n)
        self.slot_epoch.append(self.epoch)
        self.slot_voxel_keys.append(None)
        self._dirty = True
        self._invalidate()

    def extend_pose(self, slot: int, points):
        from octreelib_amd.feed import DeviceCloud

        if isinstance(points, DeviceCloud):
            # appended device-to-device behind the pose's points (the library orders the copy behind the upload)
            self.ctx.check(self.lib.octl_forest_extend_pose_device(self.handle, slot, points.ptr, points.n))
            self._device_clouds.append(points)   # (alive until the copy has been consumed by the next build)
            n_new = points.n
        else:
            pts = nat.as_points(points)
            self.ctx.check(self.lib.octl_forest_extend_pose(self.handle, slot, nat.ptr(pts), len(pts)))
            n_new = len(pts)
        self._in_place = None
        self.slot_sizes[slot] += n_new
        self._dirty = True
        self._invalidate()

    # -- build ------------------------------------------------------------------------------
    def build(self, K: int, scheme_slots=None, keep_scheme=False, max_depth=0):
        self._resolve_membership()   # (captured voxel ids refer to the voxel table this build may renumber)
        mask = None
        if scheme_slots is not None and not keep_scheme:
            mask = np.zeros(self.n_slots, dtype=np.uint8)
            mask[list(scheme_slots)] = 1
        info = nat.BuildInfo()
        try:
            self.ctx.check(
                self.lib.octl_forest_build(
                    self.handle, int(K), nat.ptr(mask), self.n_slots if mask is not None else 0,
                    1 if keep_scheme else 0, int(max_depth), C.byref(info),
                )
            )
        except (RecursionError, nat.DomainError, MemoryError, RuntimeError):
            # the library dropped the scheme (include/octreelib_hip.h): the next query rebuilds the
            # top-level voxels; the reference leaves a half-subdivided tree behind in these cases
            self.has_scheme = False
            self._dirty = True
            self.info = None
            self._invalidate()
            raise
        self.info = info
        self.n_ord = int(info.n_points)
        self._dirty = False
        self._invalidate()
        if not keep_scheme:
            self.epoch += 1
            self.has_scheme = True
        self._update_membership()

    def subdivide(self, K: int, scheme_slots=None, max_depth=0):
        self.build(K, scheme_slots, False, max_depth)

    def subdivide_callable(self, criteria, scheme_slots=None, max_depth=63):
        """Subdivision driven by arbitrary host callables (octree.py:26: a node splits when
        any(criterion(points))).  A Python callable cannot run inside a kernel: the host evaluates
        the predicates level by level on the points of the scheme octree's nodes (the union of the
        selected poses, octree_manager.py:53-61) and installs the resulting scheme; every placement
        (bucketing, partition into leaves, ordering) still happens on the device.  Cost: one
        device re-placement per tree level - meant for API completeness, not for speed.
        The order of the rows handed to a criterion is pose-major / insertion order (the
        reference's order is an artefact of an unstable argsort)."""
        self.ensure_built()
        self._resolve_membership()
        scheme = set(range(self.n_slots)) if scheme_slots is None else set(scheme_slots)
        # epochs of nodes that stay internal are inherited (cached-leaf order is history dependent)
        prev = {}
        if self.has_scheme:
            nd = self.nodes
            path = _node_paths(nd)
            vox = self.voxels
            for i in np.nonzero(nd["first_child"] >= 0)[0]:
                prev[(tuple(vox[nd["voxel"][i]]), path[i])] = int(nd["epoch"][i])
        new_epoch = self.epoch + 1
        V = len(self.voxels)
        vox_keys = [tuple(v) for v in self.voxels.tolist()]
        fc = [-1] * V
        ep = [0] * V
        node_vox = list(range(V))
        node_path = [()] * V
        frontier = list(range(V))
        empty = np.empty((0, 3), dtype=float)
        depth = 0
        while True:
            fca = np.ascontiguousarray(np.array(fc, dtype=np.int32))
            epa = np.ascontiguousarray(np.array(ep, dtype=np.int32))
            self.ctx.check(self.lib.octl_forest_set_scheme(self.handle, nat.ptr(fca), nat.ptr(epa), len(fca), new_epoch))
            info = nat.BuildInfo()
            self.ctx.check(self.lib.octl_forest_build(self.handle, 0, None, 0, 1, 0, C.byref(info)))
            self.info = info
            self.n_ord = int(info.n_points)
            self._dirty = False
            self._invalidate()
            assert np.array_equal(self.nodes["first_child"], fca), "device and host node numbering differ"
            if not frontier:
                break
            blk = self.blocks
            xyz = self.xyz
            by_node = {}
            for n, s_, st, sz in zip(blk["node"].tolist(), blk["slot"].tolist(), blk["start"].tolist(), blk["size"].tolist()):
                if s_ in scheme:
                    by_node.setdefault(n, []).append((st, sz))
            to_split = []
            for n in frontier:
                parts = by_node.get(n)
                pts = np.vstack([xyz[a : a + b] for a, b in parts]) if parts else empty
                if any([c(pts) for c in criteria]):
                    to_split.append(n)
            if not to_split:
                break
            depth += 1
            if depth > max_depth:
                raise RecursionError(f"maximum depth {max_depth} exceeded")
            frontier = []
            for n in sorted(to_split):
                fc[n] = len(fc)
                key = (vox_keys[node_vox[n]], node_path[n])
                ep[n] = prev.get(key, new_epoch)
                for j in range(8):
                    frontier.append(len(fc))
                    fc.append(-1)
                    ep.append(0)
                    node_vox.append(node_vox[n])
                    node_path.append(node_path[n] + (j,))
        self.epoch = new_epoch
        self.has_scheme = True
        self._update_membership()

    def adopt_scheme(self, other: "Forest"):
        """Make this forest's scheme the scheme of `other` (OctreeNode.subdivide_as, octree.py:34-53:
        split wherever `other` is split).  Both forests must cover the same voxels.  Nodes that
        are already internal here keep their epoch (the cached-leaf order is history dependent),
        newly split ones get a new epoch; where this forest is finer than `other` the subtree is
        merged (upstream's merge branch is defective, SURVEY 8 a8 - outside the parity domain)."""
        self.ensure_built()
        other.ensure_built()
        self._resolve_membership()
        if self.mode != other.mode or self._cube != other._cube or not np.array_equal(self.voxels, other.voxels):
            raise ValueError("subdivide_as needs two octrees over the same cube")
        ond = other.nodes
        ofc = np.ascontiguousarray(ond["first_child"], dtype=np.int32)
        prev = {}
        if self.has_scheme:
            nd = self.nodes
            path = _node_paths(nd)
            for i in np.nonzero(nd["first_child"] >= 0)[0]:
                prev[(int(nd["voxel"][i]), path[i])] = int(nd["epoch"][i])
        new_epoch = self.epoch + 1
        opath = _node_paths(ond)
        ep = np.zeros(len(ofc), dtype=np.int32)
        for i in np.nonzero(ofc >= 0)[0]:
            ep[i] = prev.get((int(ond["voxel"][i]), opath[i]), new_epoch)
        self.ctx.check(self.lib.octl_forest_set_scheme(self.handle, nat.ptr(ofc), nat.ptr(ep), len(ofc), new_epoch))
        info = nat.BuildInfo()
        self.ctx.check(self.lib.octl_forest_build(self.handle, 0, None, 0, 1, 0, C.byref(info)))
        self.info = info
        self.n_ord = int(info.n_points)
        self._dirty = False
        self._invalidate()
        self.epoch = new_epoch
        self.has_scheme = True
        self._update_membership()

    def ensure_built(self):
        """Bring the device structure up to date after insertions: a new pose inherits the
        current scheme (octree_manager.py:161-171); before any subdivide the scheme is just
        the top-level voxels."""
        if not self._dirty and self.info is not None:
            return
        if self.has_scheme:
            self.build(0, None, keep_scheme=True)
        else:
            self._resolve_membership()
            # K < 0: never split.  Not a subdivide call: the epoch does not advance.
            info = nat.BuildInfo()
            self.ctx.check(self.lib.octl_forest_build(self.handle, -1, None, 0, 0, 0, C.byref(info)))
            self.info = info
            self.n_ord = int(info.n_points)
            self._dirty = False
            self._invalidate()
            self.epoch += 1  # mirrors the library's counter of non-keep builds
            self._update_membership()

    def _update_membership(self):
        """Capture, for slots seen for the first time, the voxels they were inserted into
        (Grid.__pose_voxel_coordinates, grid.py:53,108) - as voxel ids of the table of THIS build; the keys and
        the voxel creation order are formed when someone asks or before the table can change
        (_resolve_membership): a scan that is built, fitted and dropped never pays for them."""
        for s in range(self._member_next, self.n_slots):
            n = C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_get_slot_voxels(self.handle, s, 0, None, C.byref(n)))
            vids = np.empty(n.value, dtype=np.int32)
            if n.value:
                self.ctx.check(
                    self.lib.octl_forest_get_slot_voxels(self.handle, s, n.value, nat.ptr(vids), C.byref(n))
                )
            self._member_pending.append((s, vids))
        self._member_next = self.n_slots

    def _fetch_voxels(self) -> np.ndarray:
        """The library's voxel table as it stands (no build is triggered)."""
        n = C.c_int64(0)
        self.ctx.check(self.lib.octl_forest_get_voxels(self.handle, 0, None, C.byref(n)))
        v = np.empty((n.value, 3), dtype=np.int64)
        self.ctx.check(self.lib.octl_forest_get_voxels(self.handle, n.value, nat.ptr(v), C.byref(n)))
        return v

    def _resolve_membership(self):
        if not self._member_pending:
            return
        vox = self._voxels if self._voxels is not None else self._fetch_voxels()
        for s, vids in self._member_pending:
            keys = vox[vids]
            self.slot_voxel_keys[s] = keys
            # voxels seen for the first time, in this pose's (lexicographic = np.unique) order
            codes = self._voxel_codes(keys)
            fresh = codes[~np.isin(codes, self._creation_codes)]
            if len(fresh):
                self._creation_codes = np.concatenate((self._creation_codes, fresh))
        self._member_pending = []

    def _voxel_codes(self, keys: np.ndarray) -> np.ndarray:
        """(m,3) int64 voxel corners -> one int64 per voxel (bijective: |index| < 2^20 per axis,
        include/octreelib_hip.h)."""
        q = np.asarray(keys, dtype=np.int64).reshape(-1, 3) // self._edge_int
        if self._code_origin is None:
            if len(q) == 0:
                return np.empty(0, dtype=np.int64)
            self._code_origin = q.min(axis=0)   # codes are relative to where the scene started (any int64 index)
        q = q - self._code_origin + (1 << 20)
        if len(q) and (q.min() < 0 or q.max() >= (1 << 21)):
            raise nat.DomainError("the scene has moved more than 2^20 voxels away from where it started")
        return (q[:, 0] << 42) | (q[:, 1] << 21) | q[:, 2]

    def creation_ranks(self, keys: np.ndarray) -> np.ndarray:
        """Position of every voxel of `keys` in the order voxels were first created
        (the dict order of Grid.__octrees, grid.py:56,100-109)."""
        self._resolve_membership()
        order = np.argsort(self._creation_codes, kind="stable")
        pos = np.searchsorted(self._creation_codes[order], self._voxel_codes(keys))
        return order[pos].astype(np.int64)

    # -- tables -----------------------------------------------------------------------------
    @property
    def nodes(self):
        if self._nodes is None:
            self.ensure_built()
            n = C.c_int64(0)
            self.ctx.check(
                self.lib.octl_forest_get_nodes(self.handle, 0, None, None, None, None, None, None, None, C.byref(n))
            )
            n = n.value
            t = {
                "voxel": np.empty(n, dtype=np.int32),
                "depth": np.empty(n, dtype=np.int32),
                "parent": np.empty(n, dtype=np.int32),
                "first_child": np.empty(n, dtype=np.int32),
                "corner": np.empty((n, 3), dtype=np.float64),
                "edge": np.empty(n, dtype=np.float64),
                "epoch": np.empty(n, dtype=np.int32),
            }
            m = C.c_int64(0)
            self.ctx.check(
                self.lib.octl_forest_get_nodes(
                    self.handle, n, nat.ptr(t["voxel"]), nat.ptr(t["depth"]), nat.ptr(t["parent"]),
                    nat.ptr(t["first_child"]), nat.ptr(t["corner"]), nat.ptr(t["edge"]),
                    nat.ptr(t["epoch"]), C.byref(m),
                )
            )
            self._nodes = t
        return self._nodes

    @property
    def voxels(self) -> np.ndarray:
        """(V,3) int64 voxel coordinates (= corners), lexicographic order."""
        if self._voxels is None:
            self.ensure_built()
            n = C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_get_voxels(self.handle, 0, None, C.byref(n)))
            v = np.empty((n.value, 3), dtype=np.int64)
            self.ctx.check(self.lib.octl_forest_get_voxels(self.handle, n.value, nat.ptr(v), C.byref(n)))
            self._voxels = v
        return self._voxels

    @property
    def blocks(self):
        if self._blocks is None:
            self.ensure_built()
            n = C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_get_blocks(self.handle, 0, None, None, None, None, C.byref(n)))
            n = n.value
            b = {
                "node": np.empty(n, dtype=np.int32),
                "slot": np.empty(n, dtype=np.int32),
                "start": np.empty(n, dtype=np.int64),
                "size": np.empty(n, dtype=np.int32),
            }
            m = C.c_int64(0)
            self.ctx.check(
                self.lib.octl_forest_get_blocks(
                    self.handle, n, nat.ptr(b["node"]), nat.ptr(b["slot"]), nat.ptr(b["start"]),
                    nat.ptr(b["size"]), C.byref(m),
                )
            )
            self._blocks = b
        return self._blocks

    @property
    def order(self) -> np.ndarray:
        """All non-empty (leaf, pose) blocks in the reference's listing order (slot major,
        voxel lexicographic, cached-leaf order) - computed on the device."""
        if self._order is None:
            self.ensure_built()
            nb = len(self.blocks["node"])
            e0 = np.ascontiguousarray(np.asarray(self.slot_epoch, dtype=np.int32))
            out = np.empty(nb, dtype=np.int32)
            n = C.c_int64(0)
            self.ctx.check(
                self.lib.octl_forest_reference_order(
                    self.handle, nat.ptr(e0) if self.n_slots else None, self.n_slots, nb,
                    nat.ptr(out), C.byref(n),
                )
            )
            self._order = out
        return self._order

    def slot_blocks(self, slot: int) -> np.ndarray:
        """Block ids of a slot in the reference's order (= its non-empty leaves)."""
        if self._slot_blocks is None:
            order = self.order
            slots = self.blocks["slot"][order]
            # order is slot-major: split at the slot boundaries
            bounds = np.searchsorted(slots, np.arange(self.n_slots + 1))
            self._slot_blocks = [order[bounds[s] : bounds[s + 1]] for s in range(self.n_slots)]
        return self._slot_blocks[slot]

    @property
    def xyz(self) -> np.ndarray:
        """Leaf-ordered coordinates (n,3) f64 (host copy, fetched on first use)."""
        if self._xyz is None:
            self.ensure_built()
            n = self.n_ord
            a = np.empty((n, 3), dtype=np.float64)
            if n:
                self.ctx.check(self.lib.octl_forest_get_points(self.handle, 0, n, nat.ptr(a)))
            self._xyz = a
        return self._xyz

    def gather_blocks(self, block_ids) -> np.ndarray:
        """Rows of the given blocks, concatenated in the given order: ONE device gather and one download
        (octl_forest_gather_blocks) - what the reference's get_points loops concatenate leaf by leaf
        (grid.py:234-242, octree_manager.py:121-130, octree.py:55-65).  A host copy of the whole ordered cloud that
        is there already is sliced instead."""
        ids = np.ascontiguousarray(block_ids, dtype=np.int32)
        if len(ids) == 0:
            return np.empty((0, 3), dtype=float)
        blk = self.blocks
        if self._xyz is not None:
            starts = blk["start"][ids].astype(np.int64)
            sizes = blk["size"][ids].astype(np.int64)
            offs = np.cumsum(sizes) - sizes
            idx = np.repeat(starts - offs, sizes) + np.arange(int(sizes.sum()), dtype=np.int64)
            return self._xyz[idx]
        self.ensure_built()
        n = int(blk["size"][ids].sum())
        out = np.empty((n, 3), dtype=np.float64)
        got = C.c_int64(0)
        self.ctx.check(self.lib.octl_forest_gather_blocks(self.handle, nat.ptr(ids), len(ids), n, nat.ptr(out),
                                                          C.byref(got)))
        if got.value != n:
            raise RuntimeError(f"gather_blocks: {got.value} rows on the device, {n} in the host's block table")
        return out

    @property
    def perm(self) -> np.ndarray:
        """perm[i] = index (in the concatenation of all pose clouds, slot order) of the point at
        storage position i."""
        if self._perm is None:
            self.ensure_built()
            n = C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_get_perm(self.handle, 0, None, C.byref(n)))
            p = np.empty(n.value, dtype=np.int64)
            self.ctx.check(self.lib.octl_forest_get_perm(self.handle, n.value, nat.ptr(p), C.byref(n)))
            self._perm = p
        return self._perm

    # -- counters (reference: octree.py:144-175, octree_manager.py:132-159, grid.py:343-362) ---
    def _slot_counts(self, slot: int):
        """(points, non-empty leaves) of a slot, reduced on the device (no table download)."""
        self.ensure_built()
        if self._counts is None:
            self._counts = {}
        if slot not in self._counts:
            a, b = C.c_int64(0), C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_slot_counts(self.handle, int(slot), C.byref(a), C.byref(b)))
            self._counts[slot] = (a.value, b.value)
        return self._counts[slot]

    def n_points(self, slot: int) -> int:
        return self._slot_counts(slot)[0]

    def n_leaves(self, slot: int) -> int:
        return self._slot_counts(slot)[1]

    def internal_per_voxel(self) -> np.ndarray:
        self.ensure_built()
        if self._internal is None:
            n = C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_internal_per_voxel(self.handle, 0, None, C.byref(n)))
            out = np.zeros(n.value, dtype=np.int32)
            if n.value:
                self.ctx.check(self.lib.octl_forest_internal_per_voxel(self.handle, n.value, nat.ptr(out), C.byref(n)))
            self._internal = out.astype(np.int64)
        return self._internal

    def slot_voxel_ranks(self, slot: int) -> np.ndarray:
        """Current voxel ranks of the voxels the slot was inserted into (lexicographic)."""
        self.ensure_built()
        self._resolve_membership()
        keys = self.slot_voxel_keys[slot]
        if keys is None or len(keys) == 0:
            return np.empty(0, dtype=np.int64)
        # the packed codes order like the (x, y, z) rows: one searchsorted over int64
        return np.searchsorted(self._voxel_codes(self.voxels), self._voxel_codes(keys)).astype(np.int64)

    def n_nodes(self, slot: int) -> int:
        ranks = self.slot_voxel_ranks(slot)
        if len(ranks) == 0:
            return 0
        return int((1 + 8 * self.internal_per_voxel()[ranks]).sum())

    # -- masks ------------------------------------------------------------------------------
    def apply_host_mask(self, mask: np.ndarray):
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        n = C.c_int64(0)
        self.ctx.check(self.lib.octl_forest_apply_host_mask(self.handle, nat.ptr(mask), len(mask), C.byref(n)))
        self.n_ord = n.value
        self._invalidate()

    def filter_count(self, slots, lo: int, hi: int):
        """Empty every leaf of the given slots whose point count is outside [lo, hi] (device)."""
        self.ensure_built()
        sel = np.zeros(max(self.n_slots, 1), dtype=np.uint8)
        sel[list(slots)] = 1
        n = C.c_int64(0)
        self.ctx.check(self.lib.octl_forest_filter_count(self.handle, nat.ptr(sel), self.n_slots, int(lo),
                                                         int(min(hi, (1 << 63) - 1)), C.byref(n)))
        self.n_ord = n.value
        self._invalidate()

    def set_contents(self, blk_node, blk_slot, blk_size, xyz):
        """Replace the contents of the leaves, scheme kept (map_leaf_points with a transforming function)."""
        self.ensure_built()
        node = np.ascontiguousarray(blk_node, dtype=np.int32)
        slot = np.ascontiguousarray(blk_slot, dtype=np.int32)
        size = np.ascontiguousarray(blk_size, dtype=np.int32)
        pts = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        assert int(size.sum()) == len(pts)
        self.ctx.check(self.lib.octl_forest_set_contents(self.handle, len(node), nat.ptr(node), nat.ptr(slot),
                                                         nat.ptr(size), nat.ptr(pts)))
        self.n_ord = len(pts)
        self._in_place = None
        for s in range(self.n_slots):
            self.slot_sizes[s] = int(size[slot == s].sum())
        self._invalidate()

    def apply_device_mask(self):
        # (octl_forest_apply_mask_async: the compaction is enqueued, its counts are booked when somebody asks -
        #  the reference's map_leaf_points_cuda_ransac returns nothing, grid.py:124-215)
        self.ctx.check(self.lib.octl_forest_apply_mask_async(self.handle))
        self._n_ord_pending = True
        self._invalidate()

    @property
    def n_ord(self) -> int:
        """Points in the leaf-ordered arrays (after an apply_mask whose count is still on its way: waits for it)."""
        if self._n_ord_pending:
            n = C.c_int64(0)
            self.ctx.check(self.lib.octl_forest_settle(self.handle, C.byref(n)))
            self._n_ord = n.value
            self._n_ord_pending = False
        return self._n_ord

    @n_ord.setter
    def n_ord(self, value: int):
        self._n_ord = int(value)
        self._n_ord_pending = False

    def device_mask(self) -> np.ndarray:
        n = C.c_int64(0)
        self.ctx.check(self.lib.octl_forest_get_mask(self.handle, 0, None, C.byref(n)))
        m = np.empty(n.value, dtype=np.uint8)
        self.ctx.check(self.lib.octl_forest_get_mask(self.handle, n.value, nat.ptr(m), C.byref(n)))
        return m

    # -- RANSAC -----------------------------------------------------------------------------
    def ransac_all(self, poses_per_batch: int, hypotheses: np.ndarray, threshold: float):
        self.ensure_built()
        hyp = np.ascontiguousarray(hypotheses, dtype=np.float64)
        e0 = np.ascontiguousarray(np.asarray(self.slot_epoch, dtype=np.int32))
        self.ctx.check(
            self.lib.octl_forest_ransac_all(
                self.handle, int(poses_per_batch), nat.ptr(e0), self.n_slots, nat.ptr(hyp),
                hyp.shape[0], hyp.shape[1], float(threshold),
            )
        )

    def ransac_blocks(self, block_order: np.ndarray, hypotheses: np.ndarray, threshold: float,
                      details=False):
        self.ensure_built()
        order = np.ascontiguousarray(block_order, dtype=np.int32)
        hyp = np.ascontiguousarray(hypotheses, dtype=np.float64)
        nb = len(order)
        plane = np.empty((nb, 4), dtype=np.float32) if details else None
        count = np.empty(nb, dtype=np.int32) if details else None
        index = np.empty(nb, dtype=np.int32) if details else None
        self.ctx.check(
            self.lib.octl_forest_ransac(
                self.handle, nat.ptr(order), nb, nat.ptr(hyp), hyp.shape[0], hyp.shape[1],
                float(threshold), nat.ptr(plane), nat.ptr(count), nat.ptr(index),
            )
        )
        return plane, count, index


def _node_paths(nd):
    """Child-digit path (tuple) of every scheme node."""
    n = len(nd["parent"])
    paths = [()] * n
    order = np.argsort(nd["depth"], kind="stable")
    par, fc = nd["parent"], nd["first_child"]
    for i in order.tolist():
        p = par[i]
        if p >= 0:
            paths[i] = paths[p] + (int(i - fc[p]),)
    return paths
