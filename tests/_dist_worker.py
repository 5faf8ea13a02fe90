"""Worker of tests/test_cpu_sharding.py (world_size ranks, gloo, CPU only).

Every rank draws its own part of one scene, partitions it by the owner of each point's top-level
voxel (the same hash the HIP kernel uses), exchanges the parts with an all-to-all, builds its
shard with the oracle and checks the global result against a single-process oracle build of the
whole cloud."""

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from octreelib_amd import synthetic  # noqa: E402
from octreelib_amd.distributed import voxel_indices_np, voxel_owner_np  # noqa: E402
from oracle import octree_np as onp  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_local, K, L = 6000, 24, 1.0
    dims = (4, 4, 4)
    parts = [synthetic.planar_cloud(n_local, dims, seed=1, stream=r) for r in range(world)]
    mine = parts[rank]
    owner = voxel_owner_np(voxel_indices_np(mine, L), world)
    gidx = rank * n_local + np.arange(n_local)
    order = np.argsort(owner, kind="stable")  # stable partition, like the device radix pass
    counts = np.bincount(owner, minlength=world)
    send_xyz = [torch.from_numpy(np.ascontiguousarray(mine[order][s:e]))
                for s, e in zip(np.concatenate(([0], np.cumsum(counts)[:-1])), np.cumsum(counts))]
    send_idx = [torch.from_numpy(np.ascontiguousarray(gidx[order][s:e]))
                for s, e in zip(np.concatenate(([0], np.cumsum(counts)[:-1])), np.cumsum(counts))]
    # counts exchange, then the payload
    cnt_t = torch.from_numpy(counts.astype(np.int64))
    all_cnt = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(all_cnt, cnt_t)
    recv_counts = [int(all_cnt[p][rank]) for p in range(world)]
    recv_xyz = [torch.zeros((c, 3), dtype=torch.float64) for c in recv_counts]
    recv_idx = [torch.zeros(c, dtype=torch.int64) for c in recv_counts]
    # grouped point-to-point exchange - the same shape as the RCCL ncclSend/ncclRecv group in
    # csrc/route.hip (gloo has no all_to_all)
    reqs = []
    for p in range(world):
        if p == rank:
            recv_xyz[p].copy_(send_xyz[p])
            recv_idx[p].copy_(send_idx[p])
            continue
        if send_xyz[p].numel():
            reqs.append(dist.isend(send_xyz[p], p, tag=1))
            reqs.append(dist.isend(send_idx[p], p, tag=2))
        if recv_xyz[p].numel():
            reqs.append(dist.irecv(recv_xyz[p], p, tag=1))
            reqs.append(dist.irecv(recv_idx[p], p, tag=2))
    for r in reqs:
        r.wait()
    pts = np.vstack([t.numpy() for t in recv_xyz])
    idx = np.concatenate([t.numpy() for t in recv_idx])
    # received in (source rank, original order) order == ascending global index
    assert np.all(np.diff(idx) > 0)
    assert np.all(voxel_owner_np(voxel_indices_np(pts, L), world) == rank)

    og = onp.OGrid(1)
    og.insert_points(0, pts)
    og.subdivide(K)
    local = torch.tensor([og.n_nodes(0), og.n_leaves(0), og.n_points(0)], dtype=torch.int64)
    dist.all_reduce(local)
    # leaf table keyed by (corner, edge) -> sorted GLOBAL indices, gathered on rank 0
    table = {(c.tobytes(), e.tobytes()): tuple(sorted(idx[i].tolist())) for c, e, i in og.leaf_table(0)}
    gathered = [None] * world
    dist.all_gather_object(gathered, table)
    if rank == 0:
        whole = np.vstack(parts)
        ref = onp.OGrid(1)
        ref.insert_points(0, whole)
        ref.subdivide(K)
        assert local.tolist() == [ref.n_nodes(0), ref.n_leaves(0), ref.n_points(0)], local.tolist()
        want = {(c.tobytes(), e.tobytes()): tuple(sorted(i.tolist())) for c, e, i in ref.leaf_table(0)}
        merged = {}
        for t in gathered:
            assert not (set(t) & set(merged)), "a voxel was built on two ranks"
            merged.update(t)
        assert merged == want
        print("SHARDING_OK", local.tolist())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
