#!/usr/bin/env python3
"""
bench.py - throughput of the octreelib hot path on MI355X.

One "step" = one pass of the path over one synthetic cloud that is already resident in HBM:
    insert (top-level voxel bucketing)  -> Grid.insert_points
    subdivide, count criterion len > 64 -> Grid.subdivide
    per-leaf RANSAC (1024 hyp., k = 6)  -> Grid.map_leaf_points_cuda_ransac (incl. apply_mask)
through the C ABI of liboctree_hip.so.

N = 1 (default): BASELINE config 3 - 10 M planar points over 32^3 voxels of 1 m.
N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (through
torch.distributed.run, BEFORE anything in this process touches a GPU); launched under torchrun
(WORLD_SIZE in the environment) it is one of the ranks.  Every rank holds its own part of one
larger scene, the grid is sharded by top-level voxel and one RCCL all-to-all over xGMI routes the
points to their owners inside the step (weak scaling).  Points per rank: 125 M at N = 8 - BASELINE
config 5, 10^9 points over [0,128)^3 - otherwise 10 M (`--points-per-rank` overrides; at N = 8 the
10 M/rank point of the weak-scaling series is measured as well and reported under `secondary`).

Prints ONE JSON line of under 8 KB on rank 0's stdout (contract in the task description) with `roofline`
(dominant kernel, live hipEvent timings), `roofline_build` (dominant streaming kernel of insert+subdivide)
and `cpu_baseline` (the NumPy port of the reference's algorithm, timed on this box's host on a
bounded sample of the same workload).  The full result - secondaries, per-kernel tables, notes, the
N > 1 topology and exchange blocks - goes to bench_detail.json beside this script and to stderr
(`compact_line` / `emit` below).
"""

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from octreelib_amd import _native as nat  # noqa: E402  (loading the library does not touch the GPU)
from octreelib_amd import synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector FP64, FMA counted as 2 flops
K_SPLIT = 64
H, KPTS, THRESHOLD = 1024, 6, 0.01
C5_POINTS_PER_RANK = 125_000_000   # BASELINE config 5: 10^9 points over 8 ranks
GEN_CHUNK = 10_000_000             # host-side generation granularity (bounds host memory)


def scene_dims(n_ranks: int, dense: bool):
    """Voxels of 1 m per rank: 32^3 at ~305 points per voxel (10 M points per rank, BASELINE
    config 2/3) or 64^3 at ~477 (125 M per rank: at 8 ranks the [0,128)^3 scene of config 5).
    The extent doubles along z, y, x in turn with the number of ranks."""
    s = 64 if dense else 32
    d = [s, s, s]
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


def visible_gpus() -> int:
    """Number of HIP devices, counted in a CHILD process: this one must stay clear of the GPU
    until it has started its ranks."""
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from octreelib_amd import _native as n; "
            "c = C.c_int(0); n.load().octl_device_count(C.byref(c)); print(c.value)" % ROOT)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks (children of this process,
    one per GPU, rendezvous on 127.0.0.1) and pass their exit code on."""
    have = visible_gpus()
    # (OCTL_BENCH_DEVICE: the rehearsal of the N > 1 path on a one-GPU box - every rank on that device, the
    #  collectives through the tests' RCCL stand-in; tools/rehearse.sh, tests/test_gpu_rehearsal.py)
    if have < n and os.environ.get("OCTL_BENCH_DEVICE") is None:
        print(f"bench.py: --gpus {n} needs {n} GPUs, this machine shows {have}; nothing was run",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points-per-rank", "--points", dest="points", type=int, default=None,
                    help="default: 125 M at --gpus 8 (BASELINE config 5), else 10 M")
    ap.add_argument("--cloud", choices=["planar", "uniform"], default="planar")
    ap.add_argument("--clouds", type=int, default=None,
                    help="distinct clouds resident in HBM that the timed loop rotates over (default: 3 up to 20 M points "
                         "per GPU - two draws of the scene and one moved by a voxel, which the geometry hint of the "
                         "previous step does not fit - else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary measurements")
    ap.add_argument("--no-overlap", action="store_true",
                    help="route inside the step on the compute stream instead of one step ahead on a "
                         "second context (A/B)")
    ap.add_argument("--scene", type=int, nargs=3, default=None, metavar=("X", "Y", "Z"),
                    help="scene extent in 1 m voxels (experiments)")
    ap.add_argument("--k-split", type=int, default=K_SPLIT,
                    help="count criterion len > K (experiments; the benchmarked workload is K = 64)")
    ap.add_argument("--route", action="store_true",
                    help="rehearse the multi-GPU step on one GPU: 1-rank RCCL communicator, the "
                         "local part forced through AllGather + Send/Recv")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = --points-per-rank points on every rank, the scene grows with N (default); "
                         "strong = the N = 1 workload (10 M points over 32^3 voxels, or --points-per-rank as the TOTAL) "
                         "divided among the ranks, SURVEY 8(d)")
    ap.add_argument("--workload", choices=["headline", "c4", "c5shard", "small"], default="headline",
                    help="profiling only (tools/profile_round.sh): run ONE secondary workload alone and print its "
                         "JSON - c4 = BASELINE config 4, c5shard = one rank's 125 M-point shard of config 5")
    ap.add_argument("--plan", action="store_true",
                    help="print what `--gpus N` (default 8) is going to need - host RAM, HBM, generation and run time "
                         "against the driver's 600 s - and exit; touches no GPU and starts no rank")
    ap.add_argument("--detail", default=None, metavar="PATH",
                    help="where the full result goes (default: bench_detail.json beside this script); stdout carries "
                         "one bounded line only")
    ap.add_argument("--shard-of", type=int, default=0, metavar="R",
                    help="one GPU: generate only the points rank 0 of R would own after routing (a "
                         "rank's shard of the R-rank scene, e.g. --shard-of 8 --points-per-rank 125000000 "
                         "= one rank of BASELINE config 5)")
    return ap.parse_args()


class Workload:
    """One rank's cloud in HBM + the forest it is built into + the step functions."""

    def __init__(self, ctx, rctx, rank, world, n_local, dims, cloud, k_split, route, overlap, shard_of=0,
                 n_clouds=1):
        self.ctx, self.rctx, self.lib = ctx, rctx, ctx.lib
        self.rank, self.world, self.n_local, self.dims = rank, world, n_local, dims
        self.k_split, self.route, self.overlap = k_split, route, overlap
        self.cloud_kind, self.shard_of = cloud, shard_of
        self.host_pts = None
        # The timed loop ROTATES over `n_clouds` distinct clouds that are all resident in HBM (a SLAM loop never sees
        # the same scan twice): variant 0 and 1 are different draws of the same scene (same voxel box), variant 2 is
        # a third draw moved by one voxel along x - its voxel box is not the previous build's, so the geometry hint
        # the context carries from step to step is REJECTED going into it and again coming out of it.
        self.d_clouds = [self.make_cloud(v) for v in range(max(1, n_clouds))]
        self.d_xyz = self.d_clouds[0]
        self.tick = 0
        self.rotate = True
        lib = self.lib
        self.corner = np.zeros(3)
        self.fh = C.c_void_p()
        ctx.check(lib.octl_forest_create(ctx.handle, 0, nat.ptr(self.corner), 1.0, C.byref(self.fh)))
        self.info = nat.BuildInfo()
        self.e0 = np.zeros(1, dtype=np.int32)
        self.n_alive = C.c_int64(0)
        self.async_mask = not os.environ.get("OCTL_BENCH_SYNC_APPLY_MASK")   # (A/B: the round-5 form waits per step)
        self.n_recv = C.c_int64(n_local)
        self.send_counts = np.zeros(max(world, 1), dtype=np.int64)   # points this rank sends to every rank, last routing
        self.slot = C.c_int32(0)
        np.random.seed(0)
        self.table = np.ascontiguousarray(np.random.random((H, KPTS)))

    CLOUD_VARIANTS = ["draw 0 of the scene", "draw 1 of the same scene (same voxel box)",
                      "draw 2 moved by one voxel along x (the voxel box changes: a geometry hint of the other draws is rejected, "
                      "and the context stops hinting while the box keeps changing)"]

    def make_cloud(self, variant):
        """One cloud of this rank in HBM (generated chunk by chunk on the host, uploaded in order)."""
        ctx, lib = self.ctx, self.lib
        rank, n_local, dims, cloud, shard_of = self.rank, self.n_local, self.dims, self.cloud_kind, self.shard_of
        d_xyz = C.c_void_p()
        ctx.check(lib.octl_dev_alloc(ctx.handle, n_local * 24, C.byref(d_xyz)))

        def gen(job):
            chunk_id, m = job
            stream = rank if n_local <= GEN_CHUNK else rank * 4096 + chunk_id
            stream += 1_000_000 * variant
            if shard_of > 1:
                pts = shard_cloud(m, dims, cloud, stream, shard_of)
            elif cloud == "planar":
                pts = synthetic.planar_cloud(m, dims, seed=1, stream=stream)
            elif cloud == "planar_sweep":   # the same scene in the order a rotating LiDAR delivers it (runs per voxel)
                pts = synthetic.sweep_order(synthetic.planar_cloud(m, dims, seed=1, stream=stream), seed=stream)
            elif cloud == "uniform32":   # BASELINE C2-U / C3-U: default_rng(0).random((n,3)) * 32
                pts = np.random.default_rng(variant).random((m, 3)) * 32.0
            elif cloud == "sparse":      # a terrain sheet through a 256 x 256 x 32 box + one over-dense blob
                pts = synthetic.sparse_scene(m, (256, 256, 32), seed=7 + stream)
            else:
                pts = synthetic.uniform_cloud(m, dims, seed=1000 + stream)
            if variant == 2:
                pts[:, 0] += 1.0
            return np.ascontiguousarray(pts)

        jobs = [(c, min(GEN_CHUNK, n_local - c * GEN_CHUNK)) for c in range((n_local + GEN_CHUNK - 1) // GEN_CHUNK)]
        if shard_of > 1:
            owned_voxels(dims, shard_of)   # (computed once, before the generator threads ask for it)
        done = 0
        if len(jobs) > 1:
            # large clouds are generated chunk by chunk on a few host threads (NumPy releases the GIL in its
            # generators and ufuncs); the chunks are uploaded in order as they arrive
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
                for pts in pool.map(gen, jobs):
                    ctx.check(lib.octl_dev_upload(ctx.handle, C.c_void_p(d_xyz.value + done * 24), nat.ptr(pts),
                                                  pts.nbytes))
                    done += len(pts)
        else:
            pts = gen(jobs[0])
            ctx.check(lib.octl_dev_upload(ctx.handle, d_xyz, nat.ptr(pts), pts.nbytes))
            if variant == 0:
                self.host_pts = pts
        return d_xyz

    def next_cloud(self):
        """The cloud of the next step: the resident clouds in turn."""
        if self.rotate and len(self.d_clouds) > 1:
            self.d_xyz = self.d_clouds[self.tick % len(self.d_clouds)]
        else:
            self.d_xyz = self.d_clouds[0]
        self.tick += 1

    def route_once(self):
        self.next_cloud()
        self.rctx.check(self.lib.octl_route_points(self.rctx.handle, self.d_xyz, None, self.n_local,
                                                   self.rank * self.n_local, nat.ptr(self.corner), 1.0,
                                                   C.byref(self.n_recv), nat.ptr(self.send_counts)))

    def build(self):
        self.ctx.check(self.lib.octl_forest_build(self.fh, self.k_split, None, 0, 0, 0, C.byref(self.info)))

    def compute(self):
        lib, ctx = self.lib, self.ctx
        self.build()
        ctx.check(lib.octl_forest_ransac_all(self.fh, 10, nat.ptr(self.e0), 1, nat.ptr(self.table), H, KPTS,
                                             THRESHOLD))
        if self.async_mask:
            # (the scan loop of the reference's API: map_leaf_points_cuda_ransac returns nothing, grid.py:124-215 -
            #  the count is booked when somebody asks; the next step's first kernels queue up behind this one's last)
            ctx.check(lib.octl_forest_apply_mask_async(self.fh))
        else:
            ctx.check(lib.octl_forest_apply_mask(self.fh, C.byref(self.n_alive)))

    def settle(self):
        """Book the last apply_mask's counts (points left: self.n_alive)."""
        self.ctx.check(self.lib.octl_forest_settle(self.fh, C.byref(self.n_alive)))

    def insert(self):
        lib, ctx = self.lib, self.ctx
        ctx.check(lib.octl_forest_clear(self.fh))
        if self.route:
            self.route_once()
            if self.rctx is not ctx:   # (routing runs on its own context: the routed cloud is handed over from there)
                ctx.check(lib.octl_forest_add_pose_routed_from(self.fh, self.rctx.handle, C.byref(self.slot)))
            else:
                ctx.check(lib.octl_forest_add_pose_routed(self.fh, C.byref(self.slot)))
        else:
            # (read in place: the cloud is resident in HBM, as the contract of the timed region says)
            self.next_cloud()
            ctx.check(lib.octl_forest_add_pose_adopt(self.fh, self.d_xyz, self.n_local, C.byref(self.slot)))

    def step(self):
        self.insert()
        self.compute()

    def step_build_only(self):
        self.insert()
        self.build()

    def step_from_host(self):
        lib, ctx = self.lib, self.ctx
        ctx.check(lib.octl_forest_clear(self.fh))
        ctx.check(lib.octl_forest_add_pose(self.fh, nat.ptr(self.host_pts), self.n_local, C.byref(self.slot)))
        self.compute()

    def run_pipelined(self, count):
        """`count` steps fed from the host the way a SLAM loop feeds scans: the cloud of step i+1 is uploaded
        from page-locked host memory on the copy stream (octl_dev_upload_async) while step i is built and
        fitted; the forest reads the uploaded buffer in place.  Starts and ends drained."""
        lib, ctx = self.lib, self.ctx
        nbytes = self.n_local * 24
        if not hasattr(self, "pin"):
            self.pin, self.dbuf = [], []
            for _ in range(2):
                h, d = C.c_void_p(), C.c_void_p()
                ctx.check(lib.octl_host_alloc(ctx.handle, nbytes, C.byref(h)))
                C.memmove(h, nat.ptr(self.host_pts), nbytes)   # the front end's scan buffers
                ctx.check(lib.octl_dev_alloc(ctx.handle, nbytes, C.byref(d)))
                self.pin.append(h)
                self.dbuf.append(d)
        ctx.check(lib.octl_dev_upload_async(ctx.handle, self.dbuf[0], self.pin[0], nbytes))
        for i in range(count):
            k = i & 1
            ctx.check(lib.octl_forest_clear(self.fh))
            ctx.check(lib.octl_forest_add_pose_adopt(self.fh, self.dbuf[k], self.n_local, C.byref(self.slot)))
            if i + 1 < count:
                ctx.check(lib.octl_dev_upload_async(ctx.handle, self.dbuf[1 - k], self.pin[1 - k], nbytes))
            self.compute()
        ctx.check(lib.octl_ctx_sync_uploads(ctx.handle))
        ctx.check(lib.octl_forest_clear(self.fh))   # (the forest must not keep reading a buffer close() frees)

    def run_overlapped(self, count):
        """`count` steps; the cloud of step i+1 is routed (second context, second host thread; the
        library calls release the GIL) while step i is computed.  Starts and ends drained."""
        import threading

        lib, ctx, rctx = self.lib, self.ctx, self.rctx
        routed, free, failed = threading.Event(), threading.Event(), []
        free.set()

        def router():
            try:
                for _ in range(count):
                    free.wait()
                    free.clear()
                    self.route_once()
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
                ctx.check(lib.octl_forest_clear(self.fh))
                ctx.check(lib.octl_forest_add_pose_routed_from(self.fh, rctx.handle, C.byref(self.slot)))
                # (the forest took the routed buffer over, or the library waited for its copy: the
                #  router is free for the next cloud)
                free.set()
                self.compute()
        finally:
            free.set()
            th.join()
        if failed:
            raise failed[0]

    def run(self, count):
        if count <= 0:
            return
        if self.overlap:
            self.run_overlapped(count)
        else:
            for _ in range(count):
                self.step()

    def leaf_sizes(self):
        """Block sizes of the current build (before RANSAC), for the algorithmic flop count."""
        nb = C.c_int64(0)
        self.ctx.check(self.lib.octl_forest_get_blocks(self.fh, 0, None, None, None, None, C.byref(nb)))
        sizes = np.empty(nb.value, dtype=np.int32)
        self.ctx.check(self.lib.octl_forest_get_blocks(self.fh, nb.value, None, None, None, nat.ptr(sizes),
                                                       C.byref(nb)))
        return sizes

    def close(self):
        self.lib.octl_forest_destroy(self.fh)
        for d in self.d_clouds:
            self.ctx.check(self.lib.octl_dev_free(self.ctx.handle, d))
        for d in getattr(self, "dbuf", []):
            self.ctx.check(self.lib.octl_dev_free(self.ctx.handle, d))
        for h in getattr(self, "pin", []):
            self.ctx.check(self.lib.octl_host_free(self.ctx.handle, h))


def build_path(timer_names):
    """Which build path a step took, from the library's timer names."""
    bucket = "bucket_build" in timer_names
    loop = "level_hist" in timer_names or "level_scatter" in timer_names
    general = "keygen" in timer_names
    if bucket and not general:
        return "bucket" + (" + level loop for the voxels left behind" if loop else "")
    if general:
        return "general (keygen + radix sort + level loop)" + (" after a bucket attempt" if bucket else "")
    if "prefix_scatter" in timer_names:
        return "general (a single cube: prefix partition by the first levels + level loop below them)"
    if loop:
        return "general (a single cube: one fused level-0 pass + level loop)"
    return "incremental" if "inc_place" in timer_names else "unknown"


_OWNED = {}


def owned_voxels(dims, n_ranks):
    """Linear ids (x slowest) of the voxels of the scene that rank 0 of n_ranks owns."""
    key = (tuple(dims), n_ranks)
    if key not in _OWNED:
        from octreelib_amd.distributed import owned_voxel_ids

        _OWNED[key] = owned_voxel_ids(dims, 0, n_ranks)
    return _OWNED[key]


def shard_cloud(m, dims, cloud, stream, n_ranks):
    """m points of the scene that all lie in voxels owned by rank 0 of n_ranks: what one rank holds
    after the all-to-all (uniform over the owned voxels, so the density per voxel is the scene's)."""
    vox = owned_voxels(dims, n_ranks)
    if cloud == "planar":
        return synthetic.planar_cloud(m, dims, seed=1, stream=stream, voxels=vox)
    return synthetic.uniform_cloud(m, dims, seed=1000 + stream, voxels=vox)


def kernels_per_step(timings, steps):
    return {k: {"ms_avg": v[0] / max(v[1], 1), "launches_per_step": v[1] / steps, "ms_per_step": v[0] / steps}
            for k, v in timings.items()}


def run_c4(ctx, reps=3):
    """BASELINE config 4: one OctreeManager cube, 64 poses x 1 M points handed over from the host one by one, then
    subdivide(len > 4096) over the union of all poses (octree_manager.py:36-66).  Per-kernel table from the
    library's timers (last repetition), counter bytes from the tracked profile of `bench.py --workload c4`."""
    lib = ctx.lib
    P4, n4, K4 = 64, 1_000_000, 4096
    poses4 = [np.random.default_rng(100 + p).random((n4, 3)) for p in range(P4)]
    f4 = C.c_void_p()
    ctx.check(lib.octl_forest_create(ctx.handle, 1, nat.ptr(np.zeros(3)), 1.0, C.byref(f4)))
    info4 = nat.BuildInfo()
    ins_ms, sub_ms = [], []
    for rep in range(reps):
        ctx.check(lib.octl_forest_clear(f4))
        ctx.sync()
        t1 = time.perf_counter()
        for p4 in poses4:
            ctx.check(lib.octl_forest_add_pose(f4, nat.ptr(p4), n4, None))
        ctx.sync()
        ins_ms.append((time.perf_counter() - t1) * 1e3)
        if rep == reps - 1:
            ctx.set_profiling(True)
        t1 = time.perf_counter()
        ctx.check(lib.octl_forest_build(f4, K4, None, 0, 0, 0, C.byref(info4)))
        ctx.sync()
        sub_ms.append((time.perf_counter() - t1) * 1e3)
    tm4 = ctx.timings()
    ctx.set_profiling(False)
    kern4 = kernels_per_step(tm4, 1)
    out = {
        "insert_ms": min(ins_ms[1:]), "insert_first_ms": ins_ms[0],
        "insert_GBs": P4 * n4 * 24 / (min(ins_ms[1:]) * 1e-3) / 1e9,
        "subdivide_ms": min(sub_ms[1:]), "subdivide_Mpoints_per_s": P4 * n4 / min(sub_ms[1:]) / 1e3,
        "subdivide_ms_instrumented": sub_ms[-1],
        "hbm_read_roofline_frac": 24.0 * P4 * n4 / (min(sub_ms[1:]) * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "nodes": int(info4.n_nodes), "levels": int(info4.n_levels), "path": build_path(set(tm4)),
        "roofline_build": build_summary(kern4, P4 * n4, "c4", wall_ms=min(sub_ms[1:])),
        "note": "BASELINE config 4: 64 poses x 1 M points from pageable host memory into one cube "
                "(insert: 1.5 GB over PCIe, first run incl. the growth of the store), subdivide(len > 4096) "
                "over the union of the 64 M points; the kernel table is the instrumented last repetition "
                "(hipEvents around every kernel: slower than subdivide_ms)",
    }
    lib.octl_forest_destroy(f4)
    return out


def run_no_hint(ctx, wl, timed):
    """The cross-step state the headline leans on, as a number: the same step on the SAME cloud with and without the
    geometry hint of the context's previous build (option NO_GEOM_HINT: the voxel box of the adopted cloud then comes
    from a box pass of its own, 24 B/point more) - what the first scan of a scene pays.  Measured A/B/A/B in one go,
    each leg with its own per-kernel table and host-synchronisation count, so that the difference can be read off
    kernel by kernel."""
    lib = ctx.lib

    def leg(no_hint):
        ctx.set_option("NO_GEOM_HINT", 1 if no_hint else 0)
        wl.step()
        ms_full = timed(wl.step, reps=6) * 1e3
        ms_build = timed(wl.step_build_only, reps=6) * 1e3
        c0, c1 = C.c_uint64(0), C.c_uint64(0)
        ctx.check(lib.octl_debug_host_syncs(C.byref(c0)))
        for _ in range(4):
            wl.step()
        ctx.check(lib.octl_debug_host_syncs(C.byref(c1)))
        ctx.sync()
        ctx.set_profiling(True)
        for _ in range(3):
            wl.step()
        ctx.sync()
        tm = ctx.timings()
        ctx.set_profiling(False)
        return {"ms": ms_full, "insert_subdivide_only_ms": ms_build, "host_syncs_per_step": (c1.value - c0.value) / 4.0,
                "kernels_ms_per_step": {k: round(v[0] / 3.0, 4) for k, v in sorted(tm.items())}}

    was = wl.rotate
    wl.rotate = False
    try:
        legs = [leg(False), leg(True), leg(False), leg(True)]
    finally:
        ctx.set_option("NO_GEOM_HINT", 0)
        wl.rotate = was
    wl.step()
    hint = {k: min(legs[0][k], legs[2][k]) for k in ("ms", "insert_subdivide_only_ms")}
    nohint = {k: min(legs[1][k], legs[3][k]) for k in ("ms", "insert_subdivide_only_ms")}
    names = sorted(set(legs[2]["kernels_ms_per_step"]) | set(legs[3]["kernels_ms_per_step"]))
    delta = {k: round(legs[3]["kernels_ms_per_step"].get(k, 0.0) - legs[2]["kernels_ms_per_step"].get(k, 0.0), 4)
             for k in names}
    return {"ms": nohint["ms"], "Mpoints_per_s": wl.n_local / nohint["ms"] / 1e3,
            "insert_subdivide_only_ms": nohint["insert_subdivide_only_ms"],
            "with_hint_ms": hint["ms"], "with_hint_insert_subdivide_only_ms": hint["insert_subdivide_only_ms"],
            "delta_ms": nohint["ms"] - hint["ms"],
            "delta_insert_subdivide_only_ms": nohint["insert_subdivide_only_ms"] - hint["insert_subdivide_only_ms"],
            "kernel_delta_ms_per_step": {k: v for k, v in delta.items() if abs(v) >= 0.002},
            "legs": legs,
            "note": "same step on ONE cloud, A/B/A/B (hint, no hint, hint, no hint; 6 timed steps per figure, minimum of "
                    "the two legs): option NO_GEOM_HINT = no geometry carried over from the previous build of the "
                    "context, the cloud's voxel box comes from a box pass of its own (timer `ingest`) and the "
                    "geometry from k_bucket_geom; kernel tables from fully instrumented steps"}


def run_small_scan(ctx, k_split, timed, n=100_000, steps=60):
    """The same step on one LiDAR sweep's worth of points (BASELINE config 1's size, 7^3 voxels of the planar scene):
    bound by launches and host waits, not by bytes - both are counted by the library."""
    lib = ctx.lib
    side = max(2, int(round((n / 305.0) ** (1.0 / 3.0))))
    out = {"points": n, "scene": f"{side}^3 voxels of 1 m, planar"}
    for name, n_clouds in (("one_cloud", 1), ("rotating_3_clouds", 3)):
        w = Workload(ctx, ctx, 0, 1, n, (side, side, side), "planar", k_split, False, False, n_clouds=n_clouds)
        w.run(6)
        ms = timed(w.step, reps=steps) * 1e3
        c0, c1, l0, l1 = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        ctx.check(lib.octl_debug_host_syncs(C.byref(c0)))
        ctx.check(lib.octl_debug_launches(C.byref(l0)))
        w.run(6)
        ctx.check(lib.octl_debug_host_syncs(C.byref(c1)))
        ctx.check(lib.octl_debug_launches(C.byref(l1)))
        ctx.sync()
        ctx.set_profiling(True)
        w.run(6)
        ctx.sync()
        tm = ctx.timings()
        ctx.set_profiling(False)
        out[name] = {"ms": ms, "Mpoints_per_s": n / ms / 1e3, "launches_per_step": (l1.value - l0.value) / 6.0,
                     "host_waits_per_step": (c1.value - c0.value) / 6.0, "leaves": int(w.info.n_blocks),
                     "kernels_ms_per_step_instrumented": {k: round(v[0] / 6.0, 4) for k, v in sorted(tm.items())}}
        if n_clouds == 1:
            # A/B/A: k_bucket_finish only behind the host's look at the totals (the round-5 form before the speculative
            # launch)
            ctx.set_option("NO_SPEC_FINISH", 1)
            try:
                w.run(6)
                ms_plain = timed(w.step, reps=steps) * 1e3
            finally:
                ctx.set_option("NO_SPEC_FINISH", 0)
            w.run(6)
            ms_again = timed(w.step, reps=steps) * 1e3
            out[name]["ms"] = min(ms, ms_again)
            out[name]["Mpoints_per_s"] = n / out[name]["ms"] / 1e3
            out[name]["ms_without_speculative_finish"] = ms_plain
        w.close()
    out["note"] = ("insert + subdivide(len>%d) + RANSAC(H=1024) + apply_mask of a %d-point scan, %d timed steps; host "
                   "waits poll a flag in the pinned mirror (octl_wait_mirror_flags) and are counted like "
                   "synchronisations; round 4: 0.25 ms, 30 launches" % (k_split, n, steps))
    return out


def run_c1(ctx, reps=5):
    """BASELINE config 1: a bare Octree over [0,1)^3, 100 k uniform points (default_rng(1234)), insert +
    subdivide(len > 32); the reference's own answer for exactly this input is 6601 nodes / 5748 leaves.
    CPU: the NumPy port of the reference's recursion (oracle/octree_np.py), one core.  GPU: the same through the C ABI
    (octl_forest_create(mode 1) / add_pose / build), cloud handed over from host memory as the reference's caller does."""
    from oracle import octree_np as onp

    lib = ctx.lib
    pts = np.ascontiguousarray(np.random.default_rng(1234).random((100_000, 3)))
    t0 = time.perf_counter()
    tree = onp.OTree(np.zeros(3), 1.0)
    tree.insert_points(pts)
    tree.subdivide(32)
    cpu_s = time.perf_counter() - t0
    cpu_nodes, cpu_leaves = int(tree.n_nodes), int(tree.n_leaves)
    fh = C.c_void_p()
    ctx.check(lib.octl_forest_create(ctx.handle, 1, nat.ptr(np.zeros(3)), 1.0, C.byref(fh)))
    info = nat.BuildInfo()
    d = C.c_void_p()
    ctx.check(lib.octl_dev_alloc(ctx.handle, pts.nbytes, C.byref(d)))
    ctx.check(lib.octl_dev_upload(ctx.handle, d, nat.ptr(pts), pts.nbytes))
    host_ms, dev_ms = [], []
    for rep in range(reps + 1):
        ctx.check(lib.octl_forest_clear(fh))
        ctx.sync()
        t1 = time.perf_counter()
        ctx.check(lib.octl_forest_add_pose(fh, nat.ptr(pts), len(pts), None))
        ctx.check(lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info)))
        ctx.sync()
        host_ms.append((time.perf_counter() - t1) * 1e3)
    c0, c1 = C.c_uint64(0), C.c_uint64(0)
    for rep in range(reps + 1):
        ctx.check(lib.octl_forest_clear(fh))
        ctx.sync()
        if rep == reps:
            ctx.check(lib.octl_debug_host_syncs(C.byref(c0)))
        t1 = time.perf_counter()
        ctx.check(lib.octl_forest_add_pose_adopt(fh, d, len(pts), None))
        ctx.check(lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info)))
        ctx.sync()
        dev_ms.append((time.perf_counter() - t1) * 1e3)
        if rep == reps:
            ctx.check(lib.octl_debug_host_syncs(C.byref(c1)))
    ctx.set_profiling(True)
    ctx.check(lib.octl_forest_clear(fh))
    ctx.check(lib.octl_forest_add_pose_adopt(fh, d, len(pts), None))
    ctx.check(lib.octl_forest_build(fh, 32, None, 0, 0, 0, C.byref(info)))
    ctx.sync()
    tm = ctx.timings()
    ctx.set_profiling(False)
    nb = C.c_int64(0)
    ctx.check(lib.octl_forest_get_blocks(fh, 0, None, None, None, None, C.byref(nb)))
    out = {
        "cpu_port_s": cpu_s, "cpu_port_Mpoints_per_s": len(pts) / cpu_s / 1e6, "cpu_cores": 1,
        "cpu_port_nodes": cpu_nodes, "cpu_port_leaves": cpu_leaves,
        "gpu_ms_from_host": min(host_ms[1:]), "gpu_ms_device_resident": min(dev_ms[1:]),
        "gpu_Mpoints_per_s_device_resident": len(pts) / min(dev_ms[1:]) / 1e3,
        "gpu_nodes": int(info.n_nodes), "gpu_leaves": int(nb.value), "gpu_levels": int(info.n_levels),
        "reference_known_answer": {"nodes": 6601, "leaves": 5748},
        "answers_agree": bool(cpu_nodes == 6601 and cpu_leaves == 5748 and int(info.n_nodes) == 6601 and
                              int(nb.value) == 5748),
        "host_syncs_last_build": int(c1.value - c0.value) - 1,   # (minus the ctx.sync() of the clock)
        "path": build_path(set(tm)), "kernels_ms": {k: round(v[0], 4) for k, v in sorted(tm.items())},
        "note": "BASELINE config 1 (plumbing): Octree(OctreeConfig(), [0,0,0], 1.0).insert_points(default_rng(1234)"
                ".random((100000,3))) + subdivide([len > 32]) - the reference measured in the build container: 2.08 s "
                "(SURVEY 6); cpu_port = oracle/octree_np.py on this host, one core; gpu = C ABI, best of 5",
    }
    lib.octl_forest_destroy(fh)
    ctx.check(lib.octl_dev_free(ctx.handle, d))
    return out


def run_c5_shard(ctx, k_split, timed, n_shard=C5_POINTS_PER_RANK, ranks=8, steps=3):
    """One rank's shard of BASELINE config 5 on this GPU: the 125 M points rank 0 of 8 owns after the all-to-all
    (planar scene over 128^3 voxels, hash ownership), same step as the headline; per-kernel table of its build."""
    dims = scene_dims(ranks, True)
    t0 = time.perf_counter()
    sw = Workload(ctx, ctx, 0, 1, n_shard, dims, "planar", k_split, False, False, shard_of=ranks)
    gen_s = time.perf_counter() - t0
    sw.step()
    ctx.sync()
    ms_full = timed(sw.step, reps=steps) * 1e3
    ms_build = timed(sw.step_build_only, reps=steps) * 1e3
    ctx.set_profiling(True)
    for _ in range(2):
        sw.step()
    ctx.sync()
    tm = ctx.timings()
    ctx.set_profiling(False)
    kern = kernels_per_step(tm, 2)
    out = {
        "points": n_shard, "ms": ms_full, "Mpoints_per_s": n_shard / ms_full / 1e3,
        "insert_subdivide_only_ms": ms_build, "insert_subdivide_only_Mpoints_per_s": n_shard / ms_build / 1e3,
        "hbm_read_roofline_frac_build": 24.0 * n_shard / (ms_build * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "voxels": int(sw.info.n_voxels), "leaves": int(sw.info.n_blocks), "levels": int(sw.info.n_levels),
        "path": build_path(set(tm)), "kernels": kern,
        "roofline_build": build_summary(kern, n_shard, "c5shard", wall_ms=ms_build),
        "host_generation_s": gen_s,
        "note": f"one rank's shard of BASELINE config 5 (rank 0 of {ranks}: the points it owns after routing, "
                f"scene {dims[0]}x{dims[1]}x{dims[2]} voxels of 1 m), insert + subdivide(len>{k_split}) + RANSAC + "
                f"apply_mask, {steps} timed steps; the same as `bench.py --shard-of {ranks} --points-per-rank {n_shard}`",
    }
    sw.close()
    return out


def plan(args):
    """What the N-GPU run is going to need, from the sizes of the run alone (no GPU is touched, no rank started):
    printed as one JSON object on stdout."""
    world = args.gpus if args.gpus > 1 else 8
    strong = args.scaling == "strong"
    if strong:
        n_local = (args.points if args.points else 10_000_000) // world
    else:
        n_local = args.points if args.points else (C5_POINTS_PER_RANK if world == 8 else 10_000_000)
    n_clouds = args.clouds if args.clouds else (3 if n_local <= 20_000_000 else 1)
    cloud_gb = n_local * 24 / 1e9
    # device bytes per point of one rank's step (routed store + global indices 32, send buffers 32, partition records
    # 32 per pass - two passes above 10.5 M points -, bucket staging 28, leaf-ordered arrays + block table 48, their
    # compaction targets 44, mask 1, RANSAC descriptors ~8) + the resident clouds
    two_pass = n_local > 10_500_000
    per_point = 32 + 32 + (64 if two_pass else 32) + 28 + 48 + 44 + 1 + 8
    hbm_gb = n_local * per_point / 1e9 + n_clouds * cloud_gb
    chunks = max(1, (n_local + GEN_CHUNK - 1) // GEN_CHUNK)
    chunk_gb = min(n_local, GEN_CHUNK) * 24 / 1e9
    threads = min(6, os.cpu_count() or 1)
    # a generator thread holds ~4 arrays of its chunk's size while it works; finished chunks wait for their upload
    host_gb_rank = chunk_gb * (4 * threads + min(chunks, 2 * threads))
    gen_s_per_cloud = 0.9 * chunks / max(1, min(threads, (os.cpu_count() or 1) // max(1, world)))   # ~0.9 s per 10 M-point chunk and thread
    ms_step = 4.5 * n_local / 1e7 * (1.0 if n_local <= 10_500_000 else 1.0)   # ~4.5 ms per 10 M points (RANSAC bound)
    timed_s = (args.steps + args.warmup + 12) * ms_step / 1e3
    secondaries_s = 0.0 if args.no_secondary else 24 * 4.5e-3 + 2 * (0.9 + 1.0)   # 10 M / rank + strong series: two more clouds
    import_s, comm_s = 90.0, 20.0
    wall = import_s + comm_s + n_clouds * gen_s_per_cloud + timed_s + secondaries_s
    out = {
        "plan_for": f"python bench.py --gpus {world} --steps {args.steps} --warmup {args.warmup}"
                    + (" --scaling strong" if strong else ""),
        "ranks": world, "points_per_rank": n_local, "total_points": n_local * world, "resident_clouds_per_rank": n_clouds,
        "host": {"cpu_count_here": os.cpu_count(), "generator_threads_per_rank": threads,
                 "ram_GB_per_rank_peak_estimate": round(host_gb_rank, 1),
                 "ram_GB_all_ranks_peak_estimate": round(host_gb_rank * world, 1)},
        "device": {"hbm_GB_per_rank_estimate": round(hbm_gb, 1), "hbm_GB_available": 288,
                   "bytes_per_point_estimate": per_point},
        "exchange": {"bytes_per_point": 32, "GB_sent_to_peers_per_rank_and_step": round(n_local * 32 * (world - 1) / world / 1e9, 2),
                     "MB_per_peer_message": round(n_local * 32 / world / 1e6, 1)},
        "seconds": {"import_torch_and_library_first_time": import_s, "communicator_and_first_collective": comm_s,
                    "host_generation": round(n_clouds * gen_s_per_cloud, 1), "timed_region_and_instrumented_passes": round(timed_s, 1),
                    "secondary_series": round(secondaries_s, 1), "wall_estimate": round(wall, 1), "driver_budget": 600},
        "fits_driver_budget": bool(wall < 600),
        "note": "estimates from the run's sizes and the rates measured on one-GPU boxes in rounds 3-5 (generation "
                "~0.9 s per 10 M-point chunk and thread, ~4.5 ms of step per 10 M points); no GPU was touched",
    }
    print(json.dumps(out))


def main():
    args = parse_args()
    if args.plan:
        plan(args)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on the same device, the collectives through the
    # tests' RCCL stand-in (OCTL_RCCL_LIBRARY=tests/rccl_stub/librccl_stub.so); the numbers mean nothing then
    if os.environ.get("OCTL_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["OCTL_BENCH_DEVICE"])
    dist = None
    if world > 1:
        import torch.distributed as dist  # plumbing only: rendezvous, barrier, max-reduce

        with stdout_to_stderr():  # (gloo announces its connections on stdout: the line of this run is the only thing there)
            dist.init_process_group(backend="gloo")
            dist.barrier()
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    strong = args.scaling == "strong" and world > 1
    if strong:
        # fixed TOTAL work: the N = 1 cloud (same scene, same density) cut into `world` parts of equal size
        n_total = args.points if args.points else 10_000_000
        n_local = n_total // world
        dense = n_total > 50_000_000
        scene_ranks = 1
    else:
        n_local = args.points if args.points else (C5_POINTS_PER_RANK if world == 8 else 10_000_000)
        dense = n_local > 50_000_000
        scene_ranks = args.shard_of if (world == 1 and args.shard_of > 1) else world
    dims = tuple(args.scene) if args.scene else scene_dims(scene_ranks, dense)

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
        rctx.set_option("ROUTE_SELF_SENDRECV", 1)
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

    def barrier():
        ctx.sync()
        if rctx is not ctx:
            rctx.sync()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        import torch

        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, reps=3):
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        return max_over_ranks((time.perf_counter() - t1) / reps)

    if args.workload != "headline":
        if world != 1:
            sys.exit("--workload c4 / c5shard are one-GPU profiling runs")
        if args.workload == "small":   # (secondary.small_scan_100k alone)
            res = run_small_scan(ctx, args.k_split, timed, n=args.points or 100_000)
        elif args.workload == "c4":
            res = run_c4(ctx, reps=max(3, args.steps))
        else:
            res = run_c5_shard(ctx, args.k_split, timed, steps=max(3, args.steps))
        print(json.dumps({"workload": args.workload, **res}))
        ctx.close()
        return

    n_clouds = args.clouds if args.clouds else (3 if n_local <= 20_000_000 else 1)
    n_clouds = max(1, min(n_clouds, len(Workload.CLOUD_VARIANTS)))
    wl = Workload(ctx, rctx, rank, world, n_local, dims, args.cloud, args.k_split, route, overlap,
                  shard_of=args.shard_of if world == 1 else 0, n_clouds=n_clouds)

    # ---- the timed region: W warm-up steps, then exactly K steps between barriers ----------------
    # Inside the timed region only the DOMINANT kernel (RANSAC scoring) is timed with hipEvents on its own
    # stream, as the roofline needs it; every event pair drains the pipeline for ~10 us, so the other
    # kernels' durations come from a second, fully instrumented pass of a few steps behind the timed region.
    wl.run(args.warmup)
    ctx.set_profiling(2)
    barrier()
    wl.tick = 0   # (the timed region starts with cloud 0: which cloud got how many steps is known)
    t0 = time.perf_counter()
    wl.run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    live = ctx.timings()
    ctx.set_profiling(False)
    dt = max_over_ranks(dt)
    prof_steps = max(2, min(args.steps, 5))
    ctx.set_profiling(True)
    if rctx is not ctx:
        rctx.set_profiling(True)
    wl.run(prof_steps)
    barrier()
    timings = ctx.timings()
    ctx.set_profiling(False)
    if rctx is not ctx:
        timings.update(rctx.timings())
        rctx.set_profiling(False)
    # per-step scale of the instrumented pass -> the timed region's step count (the table below divides by it)
    timings = {k: (v[0] * args.steps / prof_steps, v[1] * args.steps / prof_steps) for k, v in timings.items()}
    if "ransac" in live:
        timings["ransac"] = live["ransac"]   # the live measurement of the timed region
    wl.settle()
    info, n_alive_after = wl.info, int(wl.n_alive.value)
    leaves, nodes, levels = int(info.n_blocks), int(info.n_nodes), int(info.n_levels)

    # host <-> device round trips of one step (library-internal synchronisations, counted by the library;
    # a few extra steps outside the timed region, profiling off)
    host_syncs = launches = spec_finish = None
    if world == 1 and not route:
        c0, c1, l0, l1 = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        h0, h1, m0, m1 = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        ctx.check(lib.octl_debug_host_syncs(C.byref(c0)))
        ctx.check(lib.octl_debug_launches(C.byref(l0)))
        ctx.check(lib.octl_debug_spec_finish(C.byref(h0), C.byref(m0)))
        wl.run(2 * n_clouds)
        ctx.check(lib.octl_debug_host_syncs(C.byref(c1)))
        ctx.check(lib.octl_debug_launches(C.byref(l1)))
        ctx.check(lib.octl_debug_spec_finish(C.byref(h1), C.byref(m1)))
        host_syncs = (c1.value - c0.value) / (2.0 * n_clouds)
        launches = (l1.value - l0.value) / (2.0 * n_clouds)
        # k_bucket_finish enqueued before the host has seen the build's totals (the wait runs beside it): launches
        # that did the work / that the host had to repeat, per step
        spec_finish = {"held_per_step": (h1.value - h0.value) / (2.0 * n_clouds),
                       "missed_per_step": (m1.value - m0.value) / (2.0 * n_clouds)}

    # algorithmic flops of the RANSAC launch from the REAL leaf sizes of this rank's build (every line: N = 1,
    # one rank's shard, N > 1 - the fall-back 6 H n ignores the plane fits and made the lines incomparable)
    # With rotating clouds: the average over the clouds, weighted by the steps each one got in the timed region.
    flops, leaves_evaluated, blocks_early_possible = 0.0, 0.0, None
    per_cloud = []
    for c in range(n_clouds):
        share = len(range(c, args.steps, n_clouds)) / float(args.steps)
        wl.tick = c
        wl.insert()
        wl.build()
        sizes = wl.leaf_sizes()
        fit = sizes[sizes >= KPTS].astype(np.int64)
        fl = float(H * (20.0 * KPTS * len(fit) + 6.0 * fit.sum()) + 6.0 * fit.sum())
        flops += share * fl
        leaves_evaluated += share * len(fit)
        per_cloud.append({"cloud": Workload.CLOUD_VARIANTS[c], "steps_in_timed_region": len(range(c, args.steps, n_clouds)),
                          "leaves": int(len(sizes)), "leaves_evaluated": int(len(fit)), "voxels": int(wl.info.n_voxels)})
        del sizes, fit
    leaves_evaluated = int(round(leaves_evaluated))

    # ---- secondary figures (outside the timed region) --------------------------------------------
    secondary = {}
    if not args.no_secondary:
        if world == 1 and not route:
            if n_clouds > 1:
                wl.rotate = False
                wl.step()
                ms = timed(wl.step, reps=6) * 1e3
                wl.rotate = True
                secondary["same_cloud"] = {
                    "ms": ms, "Mpoints_per_s": n_local / ms / 1e3,
                    "note": "the headline's step on ONE cloud, inserted again every step (rounds 1-4 measured this): "
                            "the geometry hint of the previous build always fits",
                }
                wl.step()
            ms = timed(wl.step_build_only, reps=2 * n_clouds) * 1e3
            secondary["insert_subdivide_only"] = {
                "ms": ms, "Mpoints_per_s": n_local / ms / 1e3,
                "hbm_read_roofline_frac": 24.0 * n_local / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "BASELINE config 2: insert + subdivide alone (rotating over the resident clouds like the "
                        "headline), algorithmic 24 B/point",
            }
            # what a build costs when the bucket path does not apply and the level-synchronous path takes it
            # (more than 2^24 voxel keys in the box, a bucket beyond 65 535 points, a vanished voxel of a previous
            # scheme ...): the same cloud, forced down that path
            ctx.set_option("NO_BUCKET_BUILD", 1)
            try:
                wl.step_build_only()
                ms_g = timed(wl.step_build_only) * 1e3
            finally:
                ctx.set_option("NO_BUCKET_BUILD", 0)
            secondary["insert_subdivide_general_path"] = {
                "ms": ms_g, "Mpoints_per_s": n_local / ms_g / 1e3,
                "note": "insert + subdivide of the same cloud through the level-synchronous path of build.hip (keygen, "
                        "radix sort, level loop): the fallback of the bucket build",
            }
            wl.step_build_only()
            if wl.host_pts is not None:
                ms = timed(wl.step_from_host) * 1e3
                secondary["pcie_inclusive"] = {
                    "ms": ms, "Mpoints_per_s": n_local / ms / 1e3,
                    "note": "same step, cloud uploaded from pageable host memory inside the step",
                }
                wl.run_pipelined(3)
                reps = 12
                ms = timed(lambda: wl.run_pipelined(reps), reps=1) * 1e3 / reps
                secondary["pcie_pipelined"] = {
                    "ms": ms, "Mpoints_per_s": n_local / ms / 1e3,
                    "note": "same step fed from the host every step: the scan of step i+1 is uploaded from "
                            "page-locked host memory on a copy stream (octl_dev_upload_async) while step i is built "
                            "and fitted; steady state over 12 steps incl. the first, un-overlapped upload",
                }
            wl.step()  # leave the forest in the state the report describes
        if world == 1 and not route and n_local == 10_000_000 and args.cloud == "planar" and not args.shard_of:
            # the same step through the drop-in Python classes, cloud handed over from the host
            from octreelib_amd import MaxPoints
            from octreelib_amd.grid import Grid, GridConfig

            def api_step():
                grid = Grid(GridConfig(voxel_edge_length=1))
                grid.insert_points(0, wl.host_pts)
                grid.subdivide([MaxPoints(args.k_split)])
                np.random.seed(0)
                grid.map_leaf_points_cuda_ransac()
                kept = grid.n_points(0)
                grid._forest.close()
                return kept

            api_step()
            t1 = time.perf_counter()
            kept = api_step()
            ms = (time.perf_counter() - t1) * 1e3
            secondary["api_inclusive"] = {
                "ms": ms, "Mpoints_per_s": n_local / ms / 1e3, "points_after_ransac": int(kept),
                "note": "Grid(GridConfig(1)).insert_points(host cloud) + subdivide([MaxPoints(64)]) + "
                        "map_leaf_points_cuda_ransac() + n_points(), a fresh Grid per step",
            }
            # the same through the asynchronous feed of the Python package (octreelib_amd.upload_async + pinned staging)
            import octreelib_amd as oa

            stage = [oa.pinned_empty((n_local, 3)), oa.pinned_empty((n_local, 3))]
            stage[0][:] = wl.host_pts
            stage[1][:] = wl.host_pts

            def api_loop(count):
                kept = 0
                nxt = oa.upload_async(stage[0])
                for i in range(count):
                    cur = nxt
                    grid = Grid(GridConfig(voxel_edge_length=1))
                    grid.insert_points(0, cur)
                    nxt = oa.upload_async(stage[(i + 1) & 1]) if i + 1 < count else None
                    grid.subdivide([MaxPoints(args.k_split)])
                    np.random.seed(0)
                    grid.map_leaf_points_cuda_ransac()
                    kept = grid.n_points(0)
                    grid._forest.close()
                    cur.release()
                return kept

            api_loop(2)
            t1 = time.perf_counter()
            kept = api_loop(16)
            ms = (time.perf_counter() - t1) * 1e3 / 16
            secondary["api_pipelined"] = {
                "ms": ms, "Mpoints_per_s": n_local / ms / 1e3, "points_after_ransac": int(kept),
                "note": "the api_inclusive loop with scan i+1 handed over early: octreelib_amd.upload_async(pinned "
                        "staging array) -> Grid.insert_points(pose, DeviceCloud) reads the uploaded buffer in place; "
                        "16 scans (the first one's upload is not hidden: 1/16 of ~5 ms), a fresh Grid per scan",
            }
            # ... and through the two-context scan pipeline of the package (octreelib_amd.ScanPipeline): scan i+1 is
            # uploaded, inserted and subdivided on context B while scan i is still being fitted on context A
            ring = stage + [oa.pinned_empty((n_local, 3)) for _ in range(3)]   # (depth 4 + the one being drawn)
            for r in ring[2:]:
                r[:] = wl.host_pts

            def fit(grid, i):
                grid.subdivide([MaxPoints(args.k_split)])
                # (two threads: the table is handed over instead of drawn from NumPy's global generator per scan)
                grid.map_leaf_points_cuda_ransac(hypotheses=wl.table)
                return grid.n_points(0)

            with oa.ScanPipeline(2) as pipe:
                list(pipe.map((ring[i % 5] for i in range(6)), fit))
                t1 = time.perf_counter()
                kept2 = list(pipe.map((ring[i % 5] for i in range(24)), fit))
                ms = (time.perf_counter() - t1) * 1e3 / 24
            secondary["api_pipelined_2ctx"] = {
                "ms": ms, "Mpoints_per_s": n_local / ms / 1e3, "points_after_ransac": int(kept2[-1]),
                "same_result_every_scan": bool(all(k == kept for k in kept2)),
                "note": "octreelib_amd.ScanPipeline(2): 24 scans out of a ring of 5 pinned staging buffers, two worker "
                        "threads with a context each take them alternately - Grid.insert_points(DeviceCloud) + "
                        "subdivide + RANSAC + apply_mask + n_points per scan; the build of one scan overlaps the "
                        "fit of the other.  The headline stays sequential.",
            }
            del stage, ring
            # poses that arrive one at a time: 12 poses x 0.5 M points into a 16^3-voxel scheme
            # fixed by the first pose - the cost of a late pose must not grow with what is stored
            from octreelib_amd import synthetic as _syn

            lp_clouds = [_syn.planar_cloud(500_000, (16, 16, 16), seed=1, stream=p) for p in range(12)]
            lg = Grid(GridConfig(voxel_edge_length=1))
            lg.insert_points(0, lp_clouds[0])
            lg.subdivide([MaxPoints(args.k_split)])
            lg.n_leaves(0)
            lp_ms = []
            for p in range(1, 12):
                t1 = time.perf_counter()
                lg.insert_points(p, lp_clouds[p])
                lg.n_leaves(p)   # forces the placement
                lp_ms.append((time.perf_counter() - t1) * 1e3)
            t1 = time.perf_counter()
            lg.subdivide([MaxPoints(args.k_split)])
            lg.n_leaves(0)
            resub_ms = (time.perf_counter() - t1) * 1e3
            secondary["late_poses"] = {
                "insert_ms_pose_2_to_4": [round(x, 3) for x in lp_ms[1:4]],
                "insert_ms_pose_9_to_11": [round(x, 3) for x in lp_ms[8:11]],
                "resubdivide_all_12_poses_ms": resub_ms,
                "note": "Grid.insert_points(pose, 0.5 M host points) + n_leaves(pose) on a subdivided grid "
                        "(incremental placement, H2D copy included); then subdivide over the 6 M stored points",
            }
            lg._forest.close()
            del lp_clouds
            # BASELINE C2-U / C3-U: uniform cloud default_rng(0).random((10 M, 3)) * 32
            uw = Workload(ctx, ctx, 0, 1, n_local, (32, 32, 32), "uniform32", args.k_split, False, False)
            uw.step()
            ms_full = timed(uw.step) * 1e3
            ms_build = timed(uw.step_build_only) * 1e3
            secondary["uniform_scene"] = {
                "ms": ms_full, "Mpoints_per_s": n_local / ms_full / 1e3,
                "insert_subdivide_only_ms": ms_build,
                "insert_subdivide_only_Mpoints_per_s": n_local / ms_build / 1e3,
                "leaves": int(uw.info.n_blocks),
                "note": "BASELINE C2-U / C3-U: np.random.default_rng(0).random((10 M, 3)) * 32, same step",
            }
            uw.close()
            # the order real scans arrive in: the same scene emitted run by run (8 .. 64 consecutive points per voxel),
            # not shuffled point by point - the partition's common case beside its worst case
            sweep = Workload(ctx, ctx, 0, 1, n_local, dims, "planar_sweep", args.k_split, False, False)

            def build_table(w):
                w.step()
                ctx.sync()
                ctx.set_profiling(True)
                for _ in range(4):
                    w.step_build_only()
                ctx.sync()
                tm = ctx.timings()
                ctx.set_profiling(False)
                return {k: tm[k][0] / max(tm[k][1], 1) for k in ("part_hist", "part_scatter", "bucket_build", "bucket_nodes")
                        if k in tm}

            t_sweep = build_table(sweep)
            ms_full = timed(sweep.step) * 1e3
            ms_build = timed(sweep.step_build_only) * 1e3
            wl.rotate = False
            t_shuf = build_table(wl)
            wl.rotate = True
            secondary["sweep_ordered"] = {
                "ms": ms_full, "Mpoints_per_s": n_local / ms_full / 1e3,
                "insert_subdivide_only_ms": ms_build,
                "insert_subdivide_only_Mpoints_per_s": n_local / ms_build / 1e3,
                "leaves": int(sweep.info.n_blocks),
                "kernel_ms_sweep_order": {k: round(v, 4) for k, v in t_sweep.items()},
                "kernel_ms_shuffled": {k: round(v, 4) for k, v in t_shuf.items()},
                "part_scatter_ps_per_record": {"sweep_order": t_sweep.get("part_scatter", 0.0) * 1e9 / n_local,
                                               "shuffled": t_shuf.get("part_scatter", 0.0) * 1e9 / n_local},
                "part_scatter_design_GBs": {"sweep_order": 56.0 * n_local / (t_sweep["part_scatter"] * 1e-3) / 1e9
                                            if t_sweep.get("part_scatter") else None,
                                            "shuffled": 56.0 * n_local / (t_shuf["part_scatter"] * 1e-3) / 1e9
                                            if t_shuf.get("part_scatter") else None},
                "note": "octreelib_amd.synthetic.sweep_order(planar scene): the headline's points in runs of 8 .. 64 per "
                        "voxel (a rotating LiDAR's order; the reference's generator emits voxel after voxel, "
                        "test/grid/test_cuda_ransac.py:9-24) against the headline's point-by-point shuffle; same step, "
                        "same results per leaf; k_part_scatter moves 24 + 32 design bytes per record",
            }
            sweep.close()
            # a scene that is NOT dense in its bounding box (every other scene here fills all voxels of its box):
            # 10 M points on a terrain sheet through a 256 x 256 x 32 box (about 8 % of its voxels occupied) + one
            # blob at 20 x the density; which build path it takes is part of the figure
            sw = Workload(ctx, ctx, 0, 1, n_local, (256, 256, 32), "sparse", args.k_split, False, False)
            sw.step()
            ctx.sync()
            ctx.set_profiling(True)
            sw.step_build_only()
            names = set(ctx.timings())
            ctx.set_profiling(False)
            ms_full = timed(sw.step) * 1e3
            ms_build = timed(sw.step_build_only) * 1e3
            secondary["sparse_scene"] = {
                "ms": ms_full, "Mpoints_per_s": n_local / ms_full / 1e3,
                "insert_subdivide_only_ms": ms_build,
                "insert_subdivide_only_Mpoints_per_s": n_local / ms_build / 1e3,
                "voxels": int(sw.info.n_voxels), "leaves": int(sw.info.n_blocks), "levels": int(sw.info.n_levels),
                "path": build_path(names),
                "note": "octreelib_amd.synthetic.sparse_scene(10 M, (256, 256, 32)): terrain sheet ~1.6 voxels thick "
                        "+ 3 % of the points in one blob at 20 x the density; same step as the headline",
            }
            sw.close()
            secondary["no_geometry_hint"] = run_no_hint(ctx, wl, timed)
            secondary["c1_octree_100k"] = run_c1(ctx)
            secondary["small_scan_100k"] = run_small_scan(ctx, args.k_split, timed)
            secondary["c4_manager"] = run_c4(ctx)
            secondary["c5_shard"] = run_c5_shard(ctx, args.k_split, timed)
            # two independent step sequences (two contexts = two streams, two forests, two host threads): what
            # a pipeline over consecutive scans gains from overlapping the memory-bound build of one scan with
            # the VALU-bound RANSAC of another.  NOT the headline: a step there is strictly sequential.
            import threading

            ctx2 = nat.Context(local_rank)
            w2 = Workload(ctx2, ctx2, 0, 1, n_local, dims, args.cloud, args.k_split, False, False)
            w2.step()
            ctx2.sync()
            reps = 10

            def seq(w, c):
                for _ in range(reps):
                    w.step()
                c.sync()

            barrier()
            t1 = time.perf_counter()
            ths = [threading.Thread(target=seq, args=(wl, ctx)), threading.Thread(target=seq, args=(w2, ctx2))]
            [t.start() for t in ths]
            [t.join() for t in ths]
            d2 = time.perf_counter() - t1
            secondary["two_streams"] = {
                "ms_per_step": d2 / (2 * reps) * 1e3, "Mpoints_per_s": 2 * reps * n_local / d2 / 1e6,
                "note": "aggregate of two independent step sequences on two contexts of the same GPU",
            }
            w2.close()
            ctx2.close()
        if world == 8 and n_local != 10_000_000 and not args.scene:
            # the 10 M points per rank point of the weak-scaling series (the N = 1 line's per-rank load)
            w10 = Workload(ctx, rctx, rank, world, 10_000_000, scene_dims(world, False), args.cloud,
                           args.k_split, route, overlap)
            w10.run(2)
            barrier()
            t1 = time.perf_counter()
            w10.run(10)
            barrier()
            d10 = max_over_ranks(time.perf_counter() - t1)
            secondary["weak_scaling_10M_per_rank"] = {
                "ms_per_step": d10 / 10 * 1e3, "Mpoints_per_s": 10_000_000 * world * 10 / d10 / 1e6,
                "note": "same step with 10 M points per rank (the per-rank load of the N = 1 line)",
            }
            w10.close()

        if world > 1 and not strong and not args.scene:
            # the fixed-total-N series of SURVEY 8(d) in the same launch: the N = 1 cloud (10 M points over 32^3
            # voxels) divided among the ranks, same routed step
            ws = Workload(ctx, rctx, rank, world, 10_000_000 // world, scene_dims(1, False), args.cloud,
                          args.k_split, route, overlap)
            ws.run(2)
            barrier()
            t1 = time.perf_counter()
            ws.run(10)
            barrier()
            ds = max_over_ranks(time.perf_counter() - t1)
            secondary["strong_scaling_10M_total"] = {
                "ms_per_step": ds / 10 * 1e3, "Mpoints_per_s": (10_000_000 // world) * world * 10 / ds / 1e6,
                "points_per_rank": 10_000_000 // world,
                "note": "STRONG scaling: BASELINE config 3's 10 M points in total, divided among the ranks "
                        "(`--scaling strong` makes this the headline of the line)",
            }
            ws.close()

    # copy bandwidth of this box (reported beside the datasheet peak)
    bw = C.c_double(0.0)
    ctx.check(lib.octl_dev_copy_bandwidth(ctx.handle, 1 << 30, 5, C.byref(bw)))

    # ---- N > 1: what the exchange moved and how even the shards are (max / mean points per rank) ----
    exchange = None
    topology = None
    if dist is not None or args.route:
        # what the COMMUNICATOR and the devices say about the run (not the launcher's environment)
        cnt, urank, ver = C.c_int32(-1), C.c_int32(-1), C.c_int32(-1)
        rctx.check(lib.octl_comm_info(rctx.handle, C.byref(cnt), C.byref(urank), C.byref(ver)))
        bus = C.create_string_buffer(32)
        uu = (C.c_uint8 * 16)()
        ncu = C.c_int32(0)
        ctx.check(lib.octl_device_identity(ctx.handle, bus, C.cast(uu, C.c_void_p), C.byref(ncu)))
        me = {"launcher_rank": rank, "rccl_user_rank": int(urank.value), "rccl_ranks": int(cnt.value),
              "rccl_version": int(ver.value), "pci_bus_id": bus.value.decode(), "device_uuid": bytes(uu).hex(),
              "compute_units": int(ncu.value), "host": socket.gethostname(),
              "sent_to_rank_last_step": [int(v) for v in wl.send_counts[:world]]}
        allt = [me]
        if dist is not None:
            allt = [None] * world
            dist.all_gather_object(allt, me)
        devices = {(t["host"], t["pci_bus_id"], t["device_uuid"]) for t in allt}
        rehearsal = os.environ.get("OCTL_BENCH_DEVICE") is not None or world == 1
        topology = {
            "rccl_ranks": allt[0]["rccl_ranks"], "rccl_version": allt[0]["rccl_version"],
            "launcher_world_size": world, "ranks": [{k: v for k, v in t.items() if k != "sent_to_rank_last_step"} for t in allt],
            "distinct_devices": len(devices),
            "one_device_per_rank": bool(len(devices) == world),
            "rank_order_agrees": bool(all(t["rccl_user_rank"] in (-1, t["launcher_rank"]) for t in allt)),
            "alltoall_bytes_rank_to_peer_last_step": [[32 * v for v in t["sent_to_rank_last_step"]] for t in allt],
            "rehearsal_on_one_device": bool(rehearsal and world > 1),
        }
        bad = []
        if any(t["rccl_ranks"] not in (-1, world) for t in allt):
            bad.append(f"the communicator has {allt[0]['rccl_ranks']} ranks, the launcher started {world}")
        if not topology["rank_order_agrees"]:
            bad.append("a rank's number in the communicator differs from the launcher's")
        if len(devices) != world and not rehearsal:
            bad.append(f"{world} ranks share {len(devices)} devices")
        if bad:
            if rank == 0:
                print("bench.py: the run is not what the line would claim: " + "; ".join(bad), file=sys.stderr)
                print(json.dumps({"error": bad, "topology": topology}), file=sys.stderr)
            sys.exit(3)
    if dist is not None:
        mine = {"n_recv": int(wl.n_recv.value),
                "sent_to_peers": int(wl.send_counts.sum() - wl.send_counts[rank]),
                "alltoall_ms": timings.get("route_alltoall", (0.0, 0))[0] / args.steps,
                "route_kernels_ms": (timings.get("route_hist", (0.0, 0))[0] +
                                     timings.get("route_scatter", (0.0, 0))[0]) / args.steps}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        recv = [r["n_recv"] for r in allr]
        sent = [r["sent_to_peers"] for r in allr]
        a2a = max(r["alltoall_ms"] for r in allr)
        exchange = {
            "points_received_per_rank": recv,
            "imbalance_max_over_mean": max(recv) / (sum(recv) / world) if sum(recv) else None,
            "points_sent_to_peers_per_rank": sent,
            "bytes_sent_to_peers_per_rank": [32 * v for v in sent],   # 24 B coordinates + 8 B global index
            "alltoall_ms_per_step_max_over_ranks": a2a,
            "alltoall_GBs_per_rank": (32.0 * max(sent) / (a2a * 1e-3) / 1e9) if a2a > 0 else None,
            "route_kernels_ms_per_step_max_over_ranks": max(r["route_kernels_ms"] for r in allr),
            "note": "one grouped ncclSend/ncclRecv all-to-all per step (route.hip), timed with hipEvents on the "
                    "routing context's stream in the instrumented pass behind the timed region; a rank's own "
                    "part is a device copy and not counted as sent",
        }

    if rank == 0:
        total_points = n_local * world
        ms_per_step = dt / args.steps * 1e3
        value = total_points * args.steps / dt / 1e6
        kern = {k: {"ms_avg": v[0] / max(v[1], 1), "launches_per_step": v[1] / args.steps,
                    "ms_per_step": v[0] / args.steps} for k, v in timings.items()}
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        n_step = int(wl.n_recv.value) if route else n_local
        # algorithmic HBM reads: 24 B/pt to place a point, 24 B/pt more (leaf ordered) for RANSAC
        dom_bytes = 24.0 * n_step
        dom_launch_ms = kern[dom]["ms_avg"]
        achieved = dom_bytes / (dom_launch_ms * 1e-3) / 1e9
        # HBM bytes of the dominant kernel per launch from the tracked PMC profile of this command (same cloud)
        dom_traffic = None
        pt = profile_traffic("headline")
        if pt is not None and dom == "ransac" and n_step == 10_000_000 and args.cloud == "planar":
            # (two launches since round 4: blocks under 128 points - all of them on this scene - go to the
            #  128-lane instance, the 256-lane one finds its part of the list empty; both are counted)
            rows = [v for k, v in pt.items() if k.startswith("k_ransac<")]
            if rows:
                dom_traffic = sum(r["fetch_bytes_corrected"] + r["write_bytes"] for r in rows)
        device_ms = sum(k["ms_per_step"] for k in kern.values())
        ransac_ms = kern.get("ransac", {}).get("ms_per_step", 0.0)
        # algorithmic f64 flops of the RANSAC kernel: per leaf with n >= k points
        # H * (20 k + 6 n) for plane fits + scoring, + 6 n for the final mask
        valu_tflops = flops / (ransac_ms * 1e-3) / 1e12 if ransac_ms else None
        # insert + subdivide: per-kernel table (three figures each) and the dominant streaming kernel
        roofline_build = build_summary(kern, n_step, "headline")
        if dense and world == 8:
            what = "BASELINE config 5: 10^9 points, "
        elif world == 1 and n_local == 10_000_000 and not args.shard_of:
            what = "BASELINE config 3: "
        else:
            what = ""
        out = {
            "metric": "Mpoints/s insert+subdivide+RANSAC",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "host_syncs_per_step": host_syncs,
            "launches_per_step": launches,
            "speculative_finish": spec_finish,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": what + f"Grid 1 m voxels, {n_local} {args.cloud} points per GPU "
                            f"(scene {dims[0]}x{dims[1]}x{dims[2]} voxels), insert + subdivide(len>{args.k_split}) + "
                            f"map_leaf_points_cuda_ransac(H=1024, k=6, thr=0.01, poses_per_batch=10) "
                            f"incl. apply_mask"
                            + (", sharded by top-level voxel with one RCCL all-to-all" if world > 1 else "")
                            + (f", STRONG scaling: {n_local * world} points in total" if strong else "")
                            + (f", one rank's shard of a {args.shard_of}-rank scene" if args.shard_of > 1 else "")
                            + (" routed one step ahead on a second stream" if overlap else ""),
                "points_per_gpu": n_local,
                "K": args.k_split,
                "hypotheses": H,
                "leaves": leaves,
                "nodes": nodes,
                "levels": levels,
                "points_after_ransac": n_alive_after,
                "clouds": per_cloud,
                "clouds_note": f"the timed loop rotates over {n_clouds} distinct clouds resident in HBM "
                               "(leaves / nodes / points_after_ransac above: the last step's cloud)"
                               if n_clouds > 1 else "one resident cloud, inserted again every step",
            },
            "roofline": {
                "kernel": dom,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": dom_traffic,
                "traffic_profile": PROFILE_TRAFFIC + " (rocprofv3 --pmc passes of this command; "
                                   "not measured by the run that prints this line)",
                "launch_ms": dom_launch_ms,
                "algorithmic_bytes_per_launch": dom_bytes,
                "note": "the RANSAC scoring kernel is FP64-VALU bound, not HBM bound: see roofline_valu; "
                        "the streaming kernels of insert+subdivide are in roofline_build",
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
                "leaves_evaluated": leaves_evaluated,
                "executed": valu_executed(ransac_ms) if (world == 1 and n_local == 10_000_000 and
                                                          args.cloud == "planar" and not args.shard_of) else None,
                "note": "ALGORITHMIC-EQUIVALENT f64 flops per leaf with n >= k points: H*(20k + 6n) + 6n (plane fits, "
                        "scoring, final mask) over the kernel's time - it counts work the kernel never executes (a "
                        "fifth of the blocks leave after the first 64 of the 1024 hypotheses) and the f32 screen's scoring as "
                        "f64: it is not a utilisation of the machine.  `executed` is: instruction counts by type from "
                        "the tracked profile against the issue ceiling at the measured clock",
            },
            "roofline_build": roofline_build,
            "pipeline_hbm": {
                "algorithmic_bytes_per_point": 48,
                "device_ms_per_step": device_ms,
                "achieved_GBs": 48.0 * n_step / (device_ms * 1e-3) / 1e9 if device_ms else None,
                "measured_copy_GBs": bw.value / 1e9,
            },
            "kernels": kern,
            "kernels_note": "ransac: hipEvents inside the timed region (live); the others: the same step "
                            "fully instrumented for a few steps behind the timed region (an event pair costs "
                            "~10 us of pipeline, a dozen per step would be 2 % of the step)",
        }
        if topology is not None:
            out["rccl_ranks"] = topology["rccl_ranks"]
            out["topology"] = topology
        if exchange is not None:
            out["imbalance"] = exchange["imbalance_max_over_mean"]
            out["exchange"] = exchange
        if secondary:
            out["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only (the other ranks would wait)
            out["cpu_baseline"] = cpu_baseline(scene_dims(1, False), wl.table)
        emit(out, args.detail)

    wl.close()
    if route:
        lib.octl_comm_destroy(rctx.handle)
    if world > 1:
        dist.destroy_process_group()
    if rctx is not ctx:
        rctx.close()
    ctx.close()


# ---- what goes on stdout ------------------------------------------------------------------------------------------
# The driver parses ONE line of stdout.  Round 5's line had grown to 24 KB (17 secondary blocks, two per-kernel tables,
# notes) and was not parsed: the record of that round is empty.  The line is now a fixed selection of the full result,
# bounded in size; everything else goes to bench_detail.json beside this script and to stderr.
LINE_LIMIT = 8192
DETAIL_FILE = "bench_detail.json"


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _r(x, digits=6):
    """Floats of the line to six significant digits (the detail file keeps them all)."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_line(full):
    """The driver's line from the full result: the contract's keys, `roofline`, `cpu_baseline`, and the few figures the
    review reads beside them.  No notes, no per-kernel tables, no secondaries (bench_detail.json has them)."""
    cfg = full.get("config") or {}
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                        "scaling", "vs_baseline", "dtype", "data"))
    line["config"] = _pick(cfg, ("workload", "points_per_gpu", "K", "hypotheses", "leaves"))
    line["roofline"] = _pick(full.get("roofline"), ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                    "launch_ms"))
    rv = full.get("roofline_valu")
    if rv:
        line["roofline_valu"] = _pick(rv, ("achieved", "peak", "unit", "frac", "frac_no_fma"))
        # (flops the REFERENCE's algorithm asks for over the kernel's time: above 1 since round 6, when nine plane fits
        #  in ten are ruled out without being executed - not a utilisation; what runs is in the detail file)
        line["roofline_valu"]["basis"] = "algorithmic-equivalent"
    rb = full.get("roofline_build")
    if rb:
        line["roofline_build"] = _pick(rb, ("kernel", "frac", "counter_frac"))
        wb = rb.get("whole_build") or {}
        line["roofline_build"]["whole_build"] = _pick(wb, ("section8d_frac", "counter_bytes_per_point"))
    if full.get("cpu_baseline"):
        line["cpu_baseline"] = _pick(full["cpu_baseline"], ("value", "unit", "cores", "kind", "sample"))
    line.update(_pick(full, ("launches_per_step", "host_syncs_per_step")))
    if full.get("n_gpus", 1) > 1 or "rccl_ranks" in full:
        line.update(_pick(full, ("rccl_ranks", "imbalance")))
    line["detail"] = DETAIL_FILE
    line = _r(line)
    text = json.dumps(line)
    assert len(text) < LINE_LIMIT, f"bench line is {len(text)} bytes (limit {LINE_LIMIT}): the driver would not parse it"
    assert "\n" not in text
    return text


def emit(full, detail_path=None):
    """Full result -> bench_detail.json (+ stderr); the bounded line -> stdout, the only thing printed there."""
    text = compact_line(full)
    path = detail_path or os.path.join(os.path.dirname(os.path.abspath(__file__)), DETAIL_FILE)
    try:
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
            fh.write("\n")
    except OSError as e:   # (a read-only tree must not cost the run its line)
        print(f"bench.py: could not write {path}: {e}", file=sys.stderr)
    print("bench.py detail: " + json.dumps(full), file=sys.stderr)
    sys.stderr.flush()
    print(text, flush=True)


# tracked rocprofv3 --pmc profiles of the three measured workloads (tools/profile_round.sh): per-launch HBM bytes by
# kernel.  They are read beside the live timings, never measured by the run that prints the line.
def _tracked(name):
    """profiles/r06_<name> when this round's profile run has written it, else the latest earlier one."""
    here = os.path.dirname(os.path.abspath(__file__))
    for tag in ("r06", "r05", "r04"):
        p = f"profiles/{tag}_{name}"
        if os.path.exists(os.path.join(here, p)):
            return p
    return f"profiles/r06_{name}"


PROFILE_TRAFFIC = _tracked("hbm_traffic.json")
PROFILE_FILES = {
    "headline": PROFILE_TRAFFIC,
    "c4": _tracked("c4_hbm_traffic.json"),
    "c5shard": _tracked("c5shard_hbm_traffic.json"),
}


def tracked_json(name):
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), _tracked(name))) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def valu_executed(ransac_ms_live):
    """What k_ransac EXECUTES (the tracked --pmc pass of this command over the VALU instruction types, the SQ pass for
    the clock, the counting variant of the library for exits and recounts) - beside roofline_valu's algorithmic
    figure, which counts flops the kernel never executes (blocks that leave early) and counts the f32 screen's work as
    f64.  Per launch = all k_ransac instances of one step."""
    mix, sq, cnt = tracked_json("valu_mix.json"), tracked_json("sq_counters.json"), tracked_json("ransac_counts.json")
    if not mix or not sq:
        return None
    rows = {k: v for k, v in mix["kernels"].items() if k.startswith("k_ransac")}
    main = max(rows, key=lambda k: rows[k]["duration_us"]) if rows else None
    if main is None or main not in sq["kernels"]:
        return None
    tot = {t: sum(r.get(t, 0.0) for r in rows.values()) for t in
           ("add_f64", "mul_f64", "fma_f64", "trans_f64", "add_f32", "mul_f32", "fma_f32", "int32")}
    t_s = rows[main]["duration_us"] * 1e-6
    clock = sq["kernels"][main]["effective_clock_GHz"]
    all_valu = sq["kernels"][main]["wave_valu_instructions"]
    f64 = tot["add_f64"] + tot["mul_f64"] + tot["fma_f64"] + tot["trans_f64"]
    f32 = tot["add_f32"] + tot["mul_f32"] + tot["fma_f32"]
    sq_t_s = sq["kernels"][main]["duration_us"] * 1e-6
    simd_cycles = 1024 * clock * 1e9 * sq_t_s          # SIMD cycles of the SQ pass's launch (1024 SIMDs)
    out = {
        "profile": _tracked("valu_mix.json"), "kernel": main, "launch_ms_profiled": rows[main]["duration_us"] / 1e3,
        "launch_ms_live": ransac_ms_live, "effective_clock_GHz": clock,
        "wave_instructions_per_launch": {"f64": f64, "f32": f32, "int32": tot["int32"], "all_valu": all_valu, **tot},
        "f64_TFLOPs_executed": (64 * (tot["add_f64"] + tot["mul_f64"] + tot["trans_f64"]) + 128 * tot["fma_f64"]) / t_s / 1e12,
        "f32_TFLOPs_executed": (64 * (tot["add_f32"] + tot["mul_f32"]) + 128 * tot["fma_f32"]) / t_s / 1e12,
        "valu_busy_fraction": sq["kernels"][main]["valu_busy_fraction"],
        "simd_cycles_per_valu_wave_instruction": simd_cycles / all_valu if all_valu else None,
        "f64_issue_cycles_share": (4.0 * f64 / simd_cycles) if simd_cycles else None,
        "f64_share_of_valu_instructions": f64 / all_valu if all_valu else None,
        "f32_share_of_valu_instructions": f32 / all_valu if all_valu else None,
        "note": "wave-level instruction counts of the tracked profile (all k_ransac instances of a step).  "
                "effective_clock_GHz (GRBM_GUI_ACTIVE / 8 / time, the SQ pass) under-reads this kernel; in_kernel_clock_GHz "
                "(s_memtime over s_memrealtime in the counting build) is the clock it holds.  Round 6: most instructions "
                "are f32 / integer ones of the prescreen (3.2-3.7 SIMD cycles per wave instruction in isolation, "
                "profiles/r05_pkfma_probe.txt), the f64 ones of the exact plane fits (4.5-4.9 cycles) are a seventh of "
                "round 5's - the opcode ledger by phase is profiles/r06_ransac_isa.txt",
    }
    if cnt and cnt.get("in_kernel_clock_GHz"):
        out["in_kernel_clock_GHz"] = cnt["in_kernel_clock_GHz"]
    if cnt:
        out.update({k: cnt[k] for k in (
            "fraction_blocks_leaving_after_group_0", "fraction_blocks_with_prescreen", "hypotheses_prescreened_per_block",
            "survivors_per_prescreened_block", "survivor_batches_per_block", "fraction_plane_fits_executed_exactly",
            "fraction_pairs_scored", "fraction_hypotheses_recounted") if k in cnt})
        out["counts_profile"] = _tracked("ransac_counts.json")
    return out




def profile_traffic(which="headline"):
    """Per-launch HBM bytes of the tracked rocprofv3 --pmc passes of a workload (profiles/)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PROFILE_FILES[which])
    try:
        with open(path) as fh:
            return json.load(fh)["kernels"]
    except (OSError, ValueError, KeyError):
        return None


# library timer name -> prefixes of the rocprof kernel names that run under it
TIMER_KERNELS = {
    "ingest": ["k_ingest<"],
    "part_hist": ["k_part_hist<", "k_part_hist_rec", "k_geom_validate"],
    "part_scan": ["k_transpose_u32", "k_table_scan"],
    "part_scatter": ["k_part_scatter<"],
    "bucket_bounds": ["k_bucket_bounds"],
    "bucket_build": ["k_bucket_build", "k_bucket_plan", "k_bucket_chunks"],
    "bucket_nodes": ["k_bucket_finish"],
    "keygen": ["k_keygen"],
    "linkey": ["k_linkey"],
    "roots": ["k_root_tiles<", "k_make_roots"],
    "init_level0": ["k_init_level0", "k_count_scheme", "k_pre_level0", "k_cube_level0", "k_top_tree"],
    "level_prepare": ["k_split_flags", "k_compact_split"],
    "level_hist": ["k_lv_hist<"],
    "level_scatter": ["k_lv_scatter"],
    "level_children": ["k_make_children"],
    "finalize": ["k_finalize"],
    "blocks": ["k_block_tiles<", "k_block_sizes"],
    "prefix_hist": ["k_part_hist<false", "k_transpose_u32"],
    "prefix_scatter": ["k_part_scatter<"],
    "ransac": ["k_ransac<"],
}
# the kernel that runs exactly ONCE per step of a workload: dispatch counts are taken relative to it
# (k_bucket_finish: a build whose hinted geometry is rejected launches the partition kernels and the totals twice,
#  the finish once; round-4 profiles: k_bucket_totals)
PROFILE_REF = {"headline": ("k_bucket_finish<false>", "k_bucket_finish", "k_bucket_totals"),
               "c5shard": ("k_bucket_finish<false>", "k_bucket_finish", "k_bucket_totals"),
               "c4": ("k_finalize_rec",)}

# DESIGN bytes per point of the streaming kernels of insert + subdivide: what each one has to read and write
# in THIS pipeline per launch - not SURVEY 8(d)'s algorithmic 24 B/point, which is reported
# separately.  Keyed by the library's timer names.
BUILD_DESIGN_BYTES = {
    "ingest": 24,                     # the box pass of a cloud read in place (no copy)
    "part_hist": 24,                  # xyz -> bucket histogram (+ voxel box under a hinted geometry)
    "part_scatter": 24 + 32,          # xyz -> 32-byte record (xyz, voxel | child digits, index) in its bucket
    "bucket_build": 32 + 4 + 4 + 24,  # records -> leafinfo, permutation, leaf-ordered coordinates
    "bucket_nodes": 4 + 4 + 4,        # leafinfo, permutation -> position -> leaf (+ nodes, blocks: small)
    "bucket_bounds": 0,               # bucket bounds of a two-pass partition (a binary search: no stream)
    # general path
    "keygen": 24 + 16,
    "linkey": 8 + 12,
    "sort_hist": 8,
    "sort_scatter": 12 + 12,
    "roots": 8,
    "init_level0": 32 + 4 + 4 + 4,     # (single cube behind a prefix partition: record tail -> index, path, leaf)
    "level_hist": 8 + 4,
    "level_scatter": 8 + 4 + 8 + 4 + 4,
    "finalize": 4 + 4 + 24 + 4 + 24,
    "blocks": 4 + 4,
    "prefix_hist": 24,                # a big single cube: histogram over the digits of its first levels
    "prefix_scatter": 24 + 32,        # ... and the one move of the coordinates into records grouped by them
}


def counter_bytes_per_step(counters, timer, ref):
    """HBM bytes one step spends under a library timer, from a tracked profile: per-launch averages x dispatches of the
    timer's kernels, per dispatch of the workload's once-per-step kernel.  None without dispatch counts (r03 files)."""
    if not counters or ref not in counters or not counters[ref].get("dispatches"):
        return None
    tot, hit = 0.0, False
    for name, row in counters.items():
        if any(name.startswith(p) for p in TIMER_KERNELS.get(timer, [])) and row.get("dispatches"):
            tot += (row["fetch_bytes_corrected"] + row["write_bytes"]) * row["dispatches"]
            hit = True
    return tot / counters[ref]["dispatches"] if hit else None


def kernel_table(kern, n_points, which):
    """Three figures per build kernel, never one for another: (1) SURVEY 8(d)'s ALGORITHMIC 24 B/point (xyz read once
    to place a point) over the kernel's time per step - the judged definition; (2) the DESIGN bytes the kernel has to
    move in this pipeline (all its launches of a step); (3) the bytes the PMC counters saw, from the tracked profile
    of the same workload (null when there is none)."""
    counters = profile_traffic(which)
    ref = next((r for r in PROFILE_REF[which] if counters and r in counters), PROFILE_REF[which][0])
    out = {}
    for k, v in kern.items():
        if k not in BUILD_DESIGN_BYTES or v["ms_per_step"] <= 0:
            continue
        ms = v["ms_per_step"]
        cb = counter_bytes_per_step(counters, k, ref)
        design = BUILD_DESIGN_BYTES[k] * v["launches_per_step"]
        out[k] = {"ms_per_step": ms, "launches_per_step": v["launches_per_step"],
                  "section8d_GBs": 24.0 * n_points / (ms * 1e-3) / 1e9,
                  "design_bytes_per_point": design,
                  "design_GBs": design * n_points / (ms * 1e-3) / 1e9,
                  "counter_bytes_per_point": (cb / n_points) if cb else None,
                  "counter_GBs": (cb / (ms * 1e-3) / 1e9) if cb else None,
                  "counter_frac_of_8TBs": (cb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if cb else None}
    return out


def build_summary(kern, n_points, which, wall_ms=None):
    """roofline_build-style summary of a workload's insert + subdivide: per-kernel table, the dominant streaming
    kernel, the whole build against SURVEY 8(d)'s HBM-read roofline."""
    table = kernel_table(kern, n_points, which)
    if not table:
        return None
    dom = max(table, key=lambda k: table[k]["ms_per_step"])
    build_ms = sum(t["ms_per_step"] for t in table.values())
    cbs = [t["counter_bytes_per_point"] for t in table.values()]
    return {
        "kernel": dom, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "achieved": table[dom]["section8d_GBs"], "frac": table[dom]["section8d_GBs"] / HBM_PEAK_GBS,
        "definition": "SURVEY 8(d): algorithmic 24 B/point x points per launch / kernel time",
        "algorithmic_bytes_per_point": 24,
        "design_GBs": table[dom]["design_GBs"], "design_frac": table[dom]["design_GBs"] / HBM_PEAK_GBS,
        "counter_GBs": table[dom]["counter_GBs"], "counter_frac": table[dom]["counter_frac_of_8TBs"],
        "traffic": (table[dom]["counter_bytes_per_point"] * n_points) if table[dom]["counter_bytes_per_point"] else None,
        "traffic_profile": PROFILE_FILES[which],
        "whole_build": {
            "kernels_ms_per_step": build_ms, "wall_ms_per_step": wall_ms,
            "section8d_GBs": 24.0 * n_points / (build_ms * 1e-3) / 1e9,
            "section8d_frac": 24.0 * n_points / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "counter_bytes_per_point": sum(cbs) if all(c is not None for c in cbs) else None,
        },
        "all": table,
    }


if __name__ == "__main__":
    main()
