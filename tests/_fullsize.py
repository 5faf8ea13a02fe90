"""Oracle comparisons at the BASELINE configurations' real sizes (10 M, 64 M, 125 M points).

The recursive oracle (oracle/octree_np.py) builds ~0.05 M points per second and the RANSAC oracle
(oracle/ransac_np.py) ~2000 blocks per second, so at full size they are applied to SAMPLES - RANSAC
blocks of every size, whole top-level voxels - while the count-only level-synchronous oracle
(oracle/count_scheme_np.py, pinned to the same golden vectors) checks EVERY leaf and EVERY point."""

import numpy as np

from tests._util import assert_same_leaves, canon_from_list


def pick_blocks_of_every_size(sizes, per_size, rng, at_least=0):
    """block ids: up to `per_size` random blocks of every distinct size (so that the rare sizes -
    the largest leaves, the blocks below k points - are all there), topped up to `at_least`."""
    sizes = np.asarray(sizes)
    order = rng.permutation(len(sizes))
    s = sizes[order]
    o2 = np.argsort(s, kind="stable")
    s2 = s[o2]
    first = np.searchsorted(s2, np.unique(s2))
    rank = np.arange(len(s2)) - np.repeat(first, np.diff(np.append(first, len(s2))))
    sel = order[o2[rank < per_size]]
    if len(sel) < at_least:
        rest = np.setdiff1d(order, sel)
        sel = np.concatenate([sel, rest[: at_least - len(sel)]])
    return np.sort(sel)  # (storage order: a batch like the reference's, leaves concatenated in sequence)


def oracle_check_ransac_blocks(f, sel, table, thr, xyz_ord):
    """The blocks `sel` of forest f as ONE batch of the operator (cuda_ransac.py:43-155): inlier count,
    winning hypothesis, f32 plane bits and mask against oracle.ransac_np.evaluate on the same batch."""
    from oracle import ransac_np as rnp

    blk = f.blocks
    starts, sizes = blk["start"][sel], blk["size"][sel]
    plane, count, index = f.ransac_blocks(sel, table, thr, details=True)
    mask = f.device_mask()
    cloud = np.concatenate([xyz_ord[s : s + z] for s, z in zip(starts.tolist(), sizes.tolist())])
    o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes.astype(np.int32), table, thr, details=True)
    assert np.array_equal(count, o_count)
    assert np.array_equal(index, o_index)
    assert np.array_equal(plane.view(np.uint32), o_plane.view(np.uint32))
    got = np.concatenate([mask[s : s + z] for s, z in zip(starts.tolist(), sizes.tolist())]).astype(bool)
    assert np.array_equal(got, o_mask)
    return sizes, count


def voxel_points_from_chunks(chunks, chosen_coords, L=1.0):
    """chunks: iterable of (first global index, (m, 3) points).  Points whose top-level voxel
    ((p - 0) // L * L).astype(int) (grid.py:72-76) is one of chosen_coords, with their global indices."""
    chosen = {tuple(int(v) for v in c) for c in chosen_coords}
    lo = np.min(chosen_coords, axis=0)
    span = np.max(chosen_coords, axis=0) - lo + 1
    code = lambda v: ((v[:, 0] - lo[0]) * span[1] + (v[:, 1] - lo[1])) * span[2] + (v[:, 2] - lo[2])
    want = np.sort(code(np.asarray(sorted(chosen))))
    pts_out, idx_out = [], []
    for base, pts in chunks:
        v = (pts // L * L).astype(int)
        inside = ((v >= lo) & (v < lo + span)).all(axis=1)
        cand = np.nonzero(inside)[0]
        hit = cand[np.isin(code(v[cand]), want)]
        pts_out.append(pts[hit])
        idx_out.append(hit + base)
    return np.concatenate(pts_out), np.concatenate(idx_out)


def oracle_check_voxels(f, chunks, n_voxels, K, rng, slot=0, ranks=None):
    """>= n_voxels random top-level voxels of a one-pose grid forest: their points rebuilt with the
    recursive oracle (OGrid) - leaf table (corner bits, edge bits) -> original index set and the leaf
    order must be equal."""
    from oracle import octree_np as onp

    vox = f.voxels
    if ranks is None:
        ranks = np.sort(rng.choice(len(vox), size=min(n_voxels, len(vox)), replace=False))
    else:
        ranks = np.sort(np.asarray(ranks))
    pts, gidx = voxel_points_from_chunks(chunks, vox[ranks])
    og = onp.OGrid(1)
    og.insert_points(0, pts)
    og.subdivide(K)
    want = [(c, e, gidx[i]) for c, e, i in og.leaf_table(0)]
    nd, blk, perm = f.nodes, f.blocks, f.perm
    order = f.order
    order = order[blk["slot"][order] == slot]
    in_sel = np.isin(nd["voxel"][blk["node"][order]], ranks)
    got = [(nd["corner"][blk["node"][b]], nd["edge"][blk["node"][b]], perm[blk["start"][b] : blk["start"][b] + blk["size"][b]])
           for b in order[in_sel].tolist()]
    assert_same_leaves(canon_from_list(got), canon_from_list(want), ordered=True)
    assert sum(len(i) for _, _, i in got) == len(pts)
    return len(ranks), len(got)


def _key_rows(corner, edge):
    k = np.empty((len(edge), 4), dtype=np.float64)
    k[:, :3] = np.asarray(corner, dtype=np.float64) + 0.0  # (-0.0 -> +0.0)
    k[:, 3] = edge
    return k.view(np.uint64)


def match_nodes(corner_a, edge_a, corner_b, edge_b):
    """a_to_b[i] = the node of table b with the same (corner bits, edge bits) as node i of table a; the two
    tables must hold exactly the same set of nodes."""
    ka, kb = _key_rows(corner_a, edge_a), _key_rows(corner_b, edge_b)
    assert len(ka) == len(kb), f"{len(ka)} nodes, expected {len(kb)}"
    oa = np.lexsort((ka[:, 3], ka[:, 2], ka[:, 1], ka[:, 0]))
    ob = np.lexsort((kb[:, 3], kb[:, 2], kb[:, 1], kb[:, 0]))
    assert np.array_equal(ka[oa], kb[ob]), "node tables differ in (corner, edge)"
    a_to_b = np.empty(len(ka), dtype=np.int64)
    a_to_b[oa] = ob
    return a_to_b


def oracle_check_every_leaf(f, poses, K, grid, scheme_slots=None, one_slot=False):
    """The whole build against the count-only oracle: the same node set (corner bits, edge bits, leaf or
    internal), and for EVERY point of every pose the same leaf - i.e. every leaf's index set per pose.
    one_slot: `poses` are consecutive pieces of the forest's single pose (a cloud too large for one array)."""
    from oracle import count_scheme_np as cnp

    if grid:
        coords, roots = cnp.grid_roots(poses, 1)
        cs = cnp.count_scheme(poses, coords, 1, K, root_of=roots, scheme_poses=scheme_slots)
        assert np.array_equal(coords, f.voxels)  # lexicographic voxel order (np.unique(axis=0), grid.py:79-81)
    else:
        cs = cnp.count_scheme(poses, np.zeros((1, 3)), 1.0, K, scheme_poses=scheme_slots)
    nd, blk, perm = f.nodes, f.blocks, f.perm
    f2o = match_nodes(nd["corner"], nd["edge"], cs.corner, cs.edge)
    assert np.array_equal(nd["first_child"] < 0, cs.is_leaf[f2o])
    # children of an internal node are consecutive and in 4ix+2iy+iz order in both tables
    internal = np.nonzero(nd["first_child"] >= 0)[0]
    for j in (0, 3, 7):
        assert np.array_equal(f2o[nd["first_child"][internal] + j], cs.first_child[f2o[internal]] + j)
    # leaf of every stored point, in the oracle's node numbering
    leaf_at_pos = np.repeat(f2o[blk["node"]], blk["size"])
    off = np.concatenate(([0], np.cumsum([len(p) for p in poses])))
    got = np.empty(off[-1], dtype=np.int64)
    got[perm] = leaf_at_pos
    slot_at_pos = np.repeat(blk["slot"], blk["size"])
    if one_slot:
        assert not slot_at_pos.any()
    else:
        assert np.array_equal(np.searchsorted(off, perm, side="right") - 1, slot_at_pos)  # a block's points are its pose's
    for p in range(len(poses)):
        assert np.array_equal(got[off[p] : off[p + 1]], cs.leaf_of[p]), f"pose {p}: leaf assignment differs"
    return cs
