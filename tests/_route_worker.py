"""One rank PROCESS of the multi-rank routing test (tests/test_gpu_route_multirank.py).  Started through
multiprocessing's fork server - itself started by conftest.py before anything in the pytest process touched
the GPU - so no process that has initialised the GPU ever forks or execs."""

import os
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")


def clouds_of(rank, world):
    from octreelib_amd import synthetic

    a = synthetic.planar_cloud(40_000 + 777 * rank, (6, 5, 4), seed=4, stream=rank)
    a[: 50 + rank] -= 3.0   # negative voxel indices too
    b = synthetic.planar_cloud(150_000 + 1001 * rank, (6, 5, 4), seed=4, stream=10 + rank)  # grows the receive buffers
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def leaf_rows(forest, gidx_of_store, slot):
    """[(corner bytes, edge bytes, sorted global indices)] of one pose, storage order."""
    nd, blk, perm = forest.nodes, forest.blocks, forest.perm
    rows = []
    for node, sl, s, z in zip(blk["node"], blk["slot"], blk["start"], blk["size"]):
        if sl != slot:
            continue
        rows.append(((nd["corner"][node] + 0.0).tobytes(), np.float64(nd["edge"][node]).tobytes(),
                     tuple(sorted(gidx_of_store[perm[s : s + z]].tolist()))))
    return rows


def run(rank, world, conn, K):
    os.environ["OCTL_RCCL_LIBRARY"] = STUB
    os.environ.pop("OCTL_ROUTE_SELF_SENDRECV", None)
    out = {"rank": rank}
    try:
        from octreelib_amd import _native as nat
        from octreelib_amd.distributed import ShardedGrid

        def bcast(b):
            if rank == 0:
                conn.send(("uid", b))
                return b
            tag, v = conn.recv()
            assert tag == "uid"
            return v

        sg = ShardedGrid(1, rank, world, comm_broadcast=bcast, device=0)
        a, b = clouds_of(rank, world)
        # ---- 1. first pose: the empty forest takes the receive buffer over --------------------------------------
        n1 = sg.insert_points(a, index_base=1_000_000 * rank)
        g1 = sg.routed_global_indices()
        sg.subdivide(K)
        out["n1"], out["g1"] = n1, g1
        out["rows1"] = leaf_rows(sg.forest, g1, 0)
        out["counters1"] = sg.global_counters(0)          # ncclAllReduce of three counters
        # ---- 2. a domain error on ONE rank must come back on EVERY rank (collective exit), nobody hangs ------------
        bad = a[:1000].copy()
        if rank == world - 1:
            bad[17, 1] = np.nan
        try:
            sg.insert_points(bad, index_base=0)
            out["domain_error"] = None
        except nat.DomainError as e:
            out["domain_error"] = str(e)
        # ---- 3. the communicator is still healthy: a larger second pose (receive buffers grow: the grow agreement
        #         all-reduce runs), copied behind the first pose ---------------------------------------------------
        n2 = sg.insert_points(b, index_base=50_000_000 + 1_000_000 * rank)
        g2 = sg.routed_global_indices()
        sg.subdivide(K)
        store_gidx = np.concatenate([g1, g2])
        out["n2"], out["g2"] = n2, g2
        out["rows2"] = [leaf_rows(sg.forest, store_gidx, s) for s in (0, 1)]
        np.random.seed(3)
        table = np.random.random((256, 6))
        sg.ransac(table, 0.01)
        out["counters2"] = [sg.global_counters(s) for s in (0, 1)]
        sg.close()
        out["ok"] = True
    except BaseException:
        out["ok"] = False
        out["error"] = traceback.format_exc()
    conn.send(("result", out))
    conn.close()
