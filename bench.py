#!/usr/bin/env python3
"""
bench.py - throughput of the octreelib hot path on MI355X.

One "step" = one pass of the path over one synthetic cloud that is already resident in HBM:
    insert (top-level voxel bucketing)  -> Grid.insert_points
    subdivide, count criterion len > 64 -> Grid.subdivide
    per-leaf RANSAC (1024 hyp., k = 6)  -> Grid.map_leaf_points_cuda_ransac (incl. apply_mask)
through the C ABI of liboctree_hip.so.  With N > 1 ranks (one process per GPU, launched by
torch.distributed.run) every rank holds its own 10 M points of one larger scene, the grid is
sharded by top-level voxel and one RCCL all-to-all over xGMI routes the points to their owners
inside the step (weak scaling).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant
kernel, live hipEvent timings) and `cpu_baseline` (the NumPy port of the reference's algorithm,
timed on this box's host on a bounded sample of the same workload).
"""

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from octreelib_amd import _native as nat  # noqa: E402
from octreelib_amd import synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64, FMA counted as 2 flops
K_SPLIT = 64
H, KPTS, THRESHOLD = 1024, 6, 0.01


def scene_dims(n_ranks: int):
    """32768 voxels of 1 m per rank: the extent doubles along z, y, x in turn."""
    d = [32, 32, 32]
    a = 2
    r = n_ranks
    while r > 1:
        d[a] *= 2
        a = (a - 1) % 3
        r //= 2
    return tuple(d)


def cpu_baseline(dims, table, budget_points=1_000_000):
    """NumPy port of the reference's algorithm (oracle/), single thread, on a sub-box of the
    same scene holding about `budget_points` points."""
    from oracle import octree_np as onp
    from oracle import ransac_np as rnp

    side = max(2, int(round((budget_points / 305.0) ** (1.0 / 3.0))))
    box = ((0, 0, 0), (side, side, side))
    n = side ** 3 * 305
    pts = synthetic.planar_cloud(n, dims, seed=1, stream=7, box=box)
    t0 = time.perf_counter()
    og = onp.OGrid(1)
    og.insert_points(0, pts)
    og.subdivide(K_SPLIT)
    table_rows = og.leaf_table(0)
    cloud = np.vstack([pts[idx] for _, _, idx in table_rows])
    sizes = np.array([len(idx) for _, _, idx in table_rows], dtype=np.int32)
    mask = rnp.evaluate(cloud, sizes, table, THRESHOLD)
    og.apply_mask(0, mask)
    dt = time.perf_counter() - t0
    return {
        "value": n / dt / 1e6,
        "unit": "Mpoints/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} points of the same planar scene ({side}^3 voxels), insert+subdivide(K=64)+"
                  f"RANSAC(H=1024,k=6)+apply_mask, NumPy port (oracle/), {dt:.1f} s; "
                  f"host has {os.cpu_count()} cores",
    }


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when a communicator is created; the contract is ONE
    JSON line on stdout, so file descriptor 1 points at stderr while the communicator comes up."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000, help="points per rank")
    ap.add_argument("--cloud", choices=["planar", "uniform"], default="planar")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="route inside the step on the compute stream instead of one step ahead on a "
                         "second context (A/B)")
    ap.add_argument("--scene", type=int, nargs=3, default=None, metavar=("X", "Y", "Z"),
                    help="scene extent in 1 m voxels (experiments; default 32768 voxels per rank)")
    ap.add_argument("--k-split", type=int, default=K_SPLIT,
                    help="count criterion len > K (experiments; the benchmarked workload is K = 64)")
    ap.add_argument("--route", action="store_true",
                    help="rehearse the multi-GPU step on one GPU: 1-rank RCCL communicator, the "
                         "local part forced through AllGather + Send/Recv")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # plumbing only: rendezvous, barrier, max-reduce

        dist.init_process_group(backend="gloo")
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    ctx = nat.Context(local_rank)
    lib = ctx.lib
    route = world > 1 or args.route
    # Routing runs one step ahead on a second context (own stream, own RCCL communicator) driven by
    # a second host thread, so that the all-to-all of step i+1 overlaps insert/subdivide/RANSAC of
    # step i.  The timed region still contains exactly K routings and K computes (the pipeline is
    # drained before the clock starts).
    overlap = route and not args.no_overlap
    rctx = nat.Context(local_rank) if overlap else ctx
    if args.route and world == 1:
        os.environ["OCTL_ROUTE_SELF_SENDRECV"] = "1"
        buf = (C.c_uint8 * nat.UNIQUE_ID_BYTES)()
        with stdout_to_stderr():
            rctx.check(lib.octl_comm_unique_id(C.cast(buf, C.c_void_p)))
            rctx.check(lib.octl_comm_init(rctx.handle, 1, 0, C.cast(buf, C.c_void_p)))
    if world > 1:
        uid = [None]
        if rank == 0:
            buf = (C.c_uint8 * nat.UNIQUE_ID_BYTES)()
            with stdout_to_stderr():
                rctx.check(lib.octl_comm_unique_id(C.cast(buf, C.c_void_p)))
            uid[0] = bytes(buf)
        dist.broadcast_object_list(uid, src=0)
        idbuf = (C.c_uint8 * nat.UNIQUE_ID_BYTES).from_buffer_copy(uid[0])
        with stdout_to_stderr():
            rctx.check(lib.octl_comm_init(rctx.handle, world, rank, C.cast(idbuf, C.c_void_p)))
            # the first collective finishes the lazy connection set-up (and its prints)
            probe = np.zeros(1, dtype=np.int64)
            rctx.check(lib.octl_comm_allreduce_i64(rctx.handle, nat.ptr(probe), 1))

    dims = tuple(args.scene) if args.scene else scene_dims(world)
    n_local = args.points
    if args.cloud == "planar":
        pts = synthetic.planar_cloud(n_local, dims, seed=1, stream=rank)
    else:
        pts = synthetic.uniform_cloud(n_local, dims, seed=rank)
    pts = np.ascontiguousarray(pts)
    np.random.seed(0)
    table = np.ascontiguousarray(np.random.random((H, KPTS)))

    # inputs resident in HBM before the timed region
    d_xyz = C.c_void_p()
    ctx.check(lib.octl_dev_alloc(ctx.handle, pts.nbytes, C.byref(d_xyz)))
    ctx.check(lib.octl_dev_upload(ctx.handle, d_xyz, nat.ptr(pts), pts.nbytes))
    corner = np.zeros(3)
    fh = C.c_void_p()
    ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(corner), 1.0, C.byref(fh)))
    info = nat.BuildInfo()
    e0 = np.zeros(1, dtype=np.int32)
    n_alive = C.c_int64(0)
    n_recv = C.c_int64(n_local)
    slot = C.c_int32(0)

    def route_once():
        rctx.check(lib.octl_route_points(rctx.handle, d_xyz, None, n_local, rank * n_local,
                                         nat.ptr(corner), 1.0, C.byref(n_recv), None))

    def compute():
        ctx.check(lib.octl_forest_build(fh, args.k_split, None, 0, 0, 0, C.byref(info)))
        ctx.check(lib.octl_forest_ransac_all(fh, 10, nat.ptr(e0), 1, nat.ptr(table), H, KPTS, THRESHOLD))
        ctx.check(lib.octl_forest_apply_mask(fh, C.byref(n_alive)))

    def step():
        ctx.check(lib.octl_forest_clear(fh))
        if route:
            route_once()
            ctx.check(lib.octl_forest_add_pose_routed(fh, C.byref(slot)))
        else:
            ctx.check(lib.octl_forest_add_pose_device(fh, d_xyz, n_local, C.byref(slot)))
        compute()

    def run_overlapped(count):
        """`count` steps; the cloud of step i+1 is routed (second context, second host thread; the
        library calls release the GIL) while step i is computed.  Starts and ends drained."""
        import threading

        routed, free, failed = threading.Event(), threading.Event(), []
        free.set()

        def router():
            try:
                for _ in range(count):
                    free.wait()
                    free.clear()
                    route_once()
                    routed.set()
            except BaseException as exc:  # surfaces in the main thread
                failed.append(exc)
                routed.set()

        th = threading.Thread(target=router, name="route-ahead")
        th.start()
        try:
            for _ in range(count):
                routed.wait()
                routed.clear()
                if failed:
                    raise failed[0]
                ctx.check(lib.octl_forest_clear(fh))
                ctx.check(lib.octl_forest_add_pose_routed_from(fh, rctx.handle, C.byref(slot)))
                ctx.sync()   # the routed buffer has been copied into the forest: free for the next cloud
                free.set()
                compute()
        finally:
            free.set()
            th.join()
        if failed:
            raise failed[0]

    def run(count):
        if count <= 0:
            return
        if overlap:
            run_overlapped(count)
        else:
            for _ in range(count):
                step()

    def barrier():
        ctx.sync()
        if rctx is not ctx:
            rctx.sync()
        if dist is not None:
            dist.barrier()

    run(args.warmup)
    ctx.set_profiling(True)
    if rctx is not ctx:
        rctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    timings = ctx.timings()
    ctx.set_profiling(False)
    if rctx is not ctx:
        timings.update(rctx.timings())
        rctx.set_profiling(False)
    if dist is not None:
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # secondary figures (untimed region of the contract): insert+subdivide alone (BASELINE config 2)
    # and the same step fed from pageable host memory (PCIe inclusive)
    def timed(fn, reps=3):
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return (time.perf_counter() - t1) / reps

    def step_build_only():
        ctx.check(lib.octl_forest_clear(fh))
        ctx.check(lib.octl_forest_add_pose_device(fh, d_xyz, n_local, C.byref(slot)))
        ctx.check(lib.octl_forest_build(fh, args.k_split, None, 0, 0, 0, C.byref(info)))

    def step_from_host():
        ctx.check(lib.octl_forest_clear(fh))
        ctx.check(lib.octl_forest_add_pose(fh, nat.ptr(pts), n_local, C.byref(slot)))
        ctx.check(lib.octl_forest_build(fh, args.k_split, None, 0, 0, 0, C.byref(info)))
        ctx.check(lib.octl_forest_ransac_all(fh, 10, nat.ptr(e0), 1, nat.ptr(table), H, KPTS, THRESHOLD))
        ctx.check(lib.octl_forest_apply_mask(fh, C.byref(n_alive)))

    extra = {}
    if world == 1 and not route:
        extra["insert_subdivide_only_ms"] = timed(step_build_only) * 1e3
        # leaf sizes before RANSAC, for the algorithmic flop count of SURVEY.md 8(d)
        nb = C.c_int64(0)
        ctx.check(lib.octl_forest_get_blocks(fh, 0, None, None, None, None, C.byref(nb)))
        sizes = np.empty(nb.value, dtype=np.int32)
        ctx.check(lib.octl_forest_get_blocks(fh, nb.value, None, None, None, nat.ptr(sizes), C.byref(nb)))
        fit = sizes[sizes >= KPTS].astype(np.int64)
        extra["leaves_evaluated"] = int(len(fit))
        extra["ransac_flops"] = float(H * (20.0 * KPTS * len(fit) + 6.0 * fit.sum()) + 6.0 * fit.sum())
        extra["pcie_inclusive_ms"] = timed(step_from_host) * 1e3
        step()  # leave the forest in the state the report describes

    # copy bandwidth of this box (reported beside the datasheet peak)
    bw = C.c_double(0.0)
    ctx.check(lib.octl_dev_copy_bandwidth(ctx.handle, 1 << 30, 5, C.byref(bw)))

    if rank == 0:
        total_points = n_local * world
        ms_per_step = dt / args.steps * 1e3
        value = total_points * args.steps / dt / 1e6
        kern = {k: {"ms_avg": v[0] / max(v[1], 1), "launches_per_step": v[1] / args.steps,
                    "ms_per_step": v[0] / args.steps} for k, v in timings.items()}
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        n_step = int(n_recv.value) if route else n_local
        # algorithmic HBM reads: 24 B/pt to place a point, 24 B/pt more (leaf ordered) for RANSAC
        alg_bytes = {"ransac": 24.0 * n_step}
        dom_bytes = alg_bytes.get(dom, 24.0 * n_step)
        dom_launch_ms = kern[dom]["ms_avg"]
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction applied there)
        traffic = None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))["kernels"]
            t = prof.get("k_" + dom)
            if t and args.points == 10_000_000 and args.cloud == "planar" and world == 1:
                traffic = t["fetch_bytes_corrected"] + t["write_bytes"]
        except Exception:
            traffic = None
        achieved = dom_bytes / (dom_launch_ms * 1e-3) / 1e9
        device_ms = sum(k["ms_per_step"] for k in kern.values())
        ransac_ms = kern.get("ransac", {}).get("ms_per_step", 0.0)
        # algorithmic f64 flops of the RANSAC kernel: per leaf with n >= k points
        # H * (20 k + 6 n) for plane fits + scoring, + 6 n for the final mask
        flops = extra.get("ransac_flops", 6.0 * H * n_step)
        valu_tflops = flops / (ransac_ms * 1e-3) / 1e12 if ransac_ms else None
        out = {
            "metric": "Mpoints/s insert+subdivide+RANSAC",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"Grid 1 m voxels, {n_local} {args.cloud} points per GPU "
                            f"(scene {dims[0]}x{dims[1]}x{dims[2]} voxels), insert + subdivide(len>{args.k_split}) + "
                            f"map_leaf_points_cuda_ransac(H=1024, k=6, thr=0.01, poses_per_batch=10) "
                            f"incl. apply_mask"
                            + (", sharded by top-level voxel with one RCCL all-to-all" if world > 1 else "")
                            + (" routed one step ahead on a second stream" if overlap else ""),
                "points_per_gpu": n_local,
                "K": args.k_split,
                "hypotheses": H,
                "leaves": int(info.n_blocks),
                "nodes": int(info.n_nodes),
                "levels": int(info.n_levels),
                "points_after_ransac": int(n_alive.value),
            },
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "launch_ms": dom_launch_ms,
                "algorithmic_bytes_per_launch": dom_bytes,
                "note": "the RANSAC scoring kernel is FP64-VALU bound, not HBM bound: see roofline_valu",
            },
            "roofline_valu": {
                "kernel": "ransac",
                "bound": "valu_f64",
                "achieved": valu_tflops,
                "peak": FP64_VALU_PEAK_TFLOPS,
                "peak_no_fma": FP64_VALU_PEAK_TFLOPS / 2,
                "unit": "TFLOP/s",
                "frac": (valu_tflops / FP64_VALU_PEAK_TFLOPS) if valu_tflops else None,
                "frac_no_fma": (valu_tflops / (FP64_VALU_PEAK_TFLOPS / 2)) if valu_tflops else None,
                "algorithmic_flops_per_launch": flops,
                "leaves_evaluated": extra.get("leaves_evaluated"),
                "note": "algorithmic f64 flops per leaf with n >= k points: H*(20k + 6n) + 6n (plane fits, "
                        "scoring, final mask); parity mode issues separate mul/add (no FMA contraction), so "
                        "the attainable ceiling is peak/2",
            },
            "pipeline_hbm": {
                "algorithmic_bytes_per_point": 48,
                "device_ms_per_step": device_ms,
                "achieved_GBs": 48.0 * n_step / (device_ms * 1e-3) / 1e9 if device_ms else None,
                "measured_copy_GBs": bw.value / 1e9,
            },
            "kernels": kern,
        }
        if extra:
            out["secondary"] = {
                "insert_subdivide_only": {
                    "ms": extra["insert_subdivide_only_ms"],
                    "Mpoints_per_s": n_local / extra["insert_subdivide_only_ms"] / 1e3,
                },
                "pcie_inclusive": {
                    "ms": extra["pcie_inclusive_ms"],
                    "Mpoints_per_s": n_local / extra["pcie_inclusive_ms"] / 1e3,
                    "note": "same step, cloud uploaded from pageable host memory inside the step",
                },
            }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the other ranks would wait)
            out["cpu_baseline"] = cpu_baseline(dims, table)
        print(json.dumps(out))

    lib.octl_forest_destroy(fh)
    ctx.check(lib.octl_dev_free(ctx.handle, d_xyz))
    if route:
        lib.octl_comm_destroy(rctx.handle)
    if world > 1:
        dist.destroy_process_group()
    if rctx is not ctx:
        rctx.close()
    ctx.close()


if __name__ == "__main__":
    main()
