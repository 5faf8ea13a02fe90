"""
Asynchronous host feed: hand scan i+1 to the device while scan i is being built and fitted.

The reference copies every cloud to the device inside the call that needs it and waits for it
(ransac/cuda_ransac.py:57-67: cuda.to_device of the cloud, the kernel, copy_to_host).  A SLAM loop over scans
can hide that copy - 240 MB, about 4.9 ms over PCIe Gen5 for 10 M points - behind the compute of the
previous scan:

    buf = [pinned_empty((n, 3)), pinned_empty((n, 3))]        # page-locked staging, filled by the front end
    nxt = upload_async(buf[0])                                 # DMA on the copy stream, returns at once
    for i in range(n_scans):
        cur, nxt = nxt, (upload_async(buf[(i + 1) % 2]) if i + 1 < n_scans else None)
        grid = Grid(GridConfig(voxel_edge_length=1))
        grid.insert_points(0, cur)                             # read in place: no copy, no host wait
        grid.subdivide([MaxPoints(64)])
        grid.map_leaf_points_cuda_ransac()
        ...
        cur.release()

Plain NumPy arrays work too (the copy then blocks the caller while HIP stages the pageable memory; it still
overlaps device work that was enqueued before).
"""

import ctypes as C
import weakref

import numpy as np

from octreelib_amd import _native as nat

__all__ = ["DeviceCloud", "ScanPipeline", "pinned_empty", "upload_async"]


def pinned_empty(shape, dtype=np.float64, ctx=None) -> np.ndarray:
    """np.empty in page-locked host memory (hipHostMalloc): uploads out of it are DMA transfers the host does
    not wait for.  The memory is released when the array (and every view of it) is gone."""
    ctx = ctx if ctx is not None else nat.get_context()
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    ctx.check(ctx.lib.octl_host_alloc(ctx.handle, max(nbytes, 1), C.byref(p)))
    buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
    weakref.finalize(buf, _free_pinned, ctx, p.value)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def _free_pinned(ctx, address):
    try:
        if getattr(ctx, "handle", None) is not None and ctx.handle.value:
            ctx.lib.octl_host_free(ctx.handle, C.c_void_p(address))
    except Exception:  # pragma: no cover - interpreter shutdown
        pass


class DeviceCloud:
    """An (n, 3) f64 cloud on its way into (or already in) device memory.  Grid.insert_points /
    OctreeManager.insert_points / Octree.insert_points take it in place of the host array: the first pose of
    an empty grid reads the device buffer in place (octl_forest_add_pose_adopt), any other pose copies it
    device-to-device.  The object owns the device buffer: keep it alive while a grid reads it in place."""

    def __init__(self, points, ctx=None):
        self.ctx = ctx if ctx is not None else nat.get_context()
        pts = nat.as_points(points)
        self.n = len(pts)
        self._host = pts  # must stay unchanged until the upload has finished
        self._readers = []  # weak references to the forests that read the buffer in place
        self.ptr = C.c_void_p()
        self.ctx.check(self.ctx.lib.octl_dev_alloc(self.ctx.handle, max(pts.nbytes, 16), C.byref(self.ptr)))
        try:
            self.ctx.check(self.ctx.lib.octl_dev_upload_async(self.ctx.handle, self.ptr, nat.ptr(pts), pts.nbytes))
        except Exception:
            self.release()
            raise

    def wait(self):
        """Host waits until the upload has finished (the host array may be overwritten afterwards)."""
        self.ctx.check(self.ctx.lib.octl_ctx_sync_uploads(self.ctx.handle))
        self._host = None

    def release(self):
        """Give the device buffer back.  Refused while a grid still reads it in place (close the grid first)."""
        for ref in getattr(self, "_readers", []):
            f = ref()
            if f is not None and getattr(f, "handle", None) is not None and f.handle.value and f.reads_in_place(self):
                raise RuntimeError("DeviceCloud.release(): a Grid / Octree still reads this buffer in place; "
                                   "close it (or insert more points into it) first")
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            self.ctx.lib.octl_dev_free(self.ctx.handle, self.ptr)  # (waits for both streams)
            self.ptr = C.c_void_p()
        self._host = None

    def __len__(self):
        return self.n

    def __del__(self):  # pragma: no cover
        try:
            self._readers = []
            self.release()
        except Exception:
            pass


def upload_async(points, ctx=None) -> DeviceCloud:
    """Start the host-to-device copy of a cloud on the context's copy stream and return at once."""
    return DeviceCloud(points, ctx)


class _StagedCloud(DeviceCloud):
    """A scan in one of the pipeline's device staging buffers: fully uploaded before a worker sees it; release()
    hands the buffer back to the uploader instead of freeing it."""

    def __init__(self, ctx, ptr, n, give_back):
        self.ctx, self.ptr, self.n = ctx, ptr, int(n)
        self._host = None
        self._readers = []
        self._give_back = give_back

    def wait(self):
        pass

    def release(self):
        if self._give_back is not None:
            give_back, self._give_back = self._give_back, None
            give_back()

    def __del__(self):  # pragma: no cover
        pass


class ScanPipeline:
    """Scans through the drop-in classes on TWO device contexts, so that consecutive scans overlap on the GPU.

    One scan is strictly sequential - upload, insert, subdivide, RANSAC, apply_mask - and its phases load different
    parts of the machine: the upload is PCIe (4.4 ms of a 10 M-point scan), the build is memory bound (~0.9 ms),
    the RANSAC scoring VALU bound (~4 ms).  The reference runs scans one after the other and waits for every copy
    (ransac/cuda_ransac.py:57-80).  Here
      * ONE uploader thread with a context of its own copies the scans, one at a time, into a small ring of device
        staging buffers (PCIe is a serial resource: two uploads at once only delay both);
      * `n_contexts` worker threads, each with its own context (stream, scratch, buffer pool), take the uploaded
        scans alternately and call `fn(grid, index)` with a fresh Grid that holds the scan as pose 0 (read in place
        from the staging buffer): while worker A fits scan i, worker B builds and fits scan i+1 and the uploader
        copies scan i+2.  The library calls release the GIL, so the threads' work interleaves on the device.
    Every scan's result is exactly what the sequential loop gives (the scans share nothing); results come back in
    submission order.

        pipe = ScanPipeline()
        def fit(grid, i):
            grid.subdivide([MaxPoints(64)])
            grid.map_leaf_points_cuda_ransac(hypotheses=table)   # (NumPy's global generator is not per thread)
            return grid.n_points(0)
        for kept in pipe.map(scans, fit):      # scans: (n, 3) arrays, ideally in pinned_empty() memory
            ...
        pipe.close()
    """

    def __init__(self, n_contexts: int = 2, voxel_edge_length=1, device=None, staging_buffers=None):
        import queue
        import threading

        if n_contexts < 1:
            raise ValueError("n_contexts must be at least 1")
        self._device = nat.default_device() if device is None else int(device)
        self._edge = voxel_edge_length
        self._n_staging = n_contexts + 1 if staging_buffers is None else max(1, int(staging_buffers))
        self._inbox = queue.Queue()                                   # (points, fn, index, future) -> the uploader
        self._jobs = [queue.Queue() for _ in range(n_contexts)]       # (job, staged cloud) -> the workers
        self._free = queue.Queue()                                    # staging buffers handed back by the workers
        self._threads = []
        self._next = 0
        self._closed = False
        up = threading.Thread(target=self._uploader, name="octl-scan-upload", daemon=True)
        up.start()
        self._threads.append(up)
        for w in range(n_contexts):
            t = threading.Thread(target=self._worker, args=(w,), name=f"octl-scan-{w}", daemon=True)
            t.start()
            self._threads.append(t)

    def _uploader(self):
        ctx = nat.Context(self._device)
        lib = ctx.lib
        bufs = []                     # [pointer, capacity in bytes]
        for _ in range(self._n_staging):
            bufs.append([C.c_void_p(), 0])
            self._free.put(len(bufs) - 1)
        try:
            while True:
                job = self._inbox.get()
                if job is None:
                    return
                points, fn, index, fut = job
                if not fut.set_running_or_notify_cancel():
                    continue
                b = self._free.get()  # (back-pressure: at most n_staging scans are uploaded ahead of their fit)
                try:
                    if isinstance(points, DeviceCloud):
                        cloud = points
                        self._free.put(b)
                    else:
                        pts = nat.as_points(points)
                        if pts.nbytes > bufs[b][1]:
                            if bufs[b][0].value:
                                ctx.check(lib.octl_dev_free(ctx.handle, bufs[b][0]))
                            bufs[b] = [C.c_void_p(), 0]
                            ctx.check(lib.octl_dev_alloc(ctx.handle, max(pts.nbytes, 16), C.byref(bufs[b][0])))
                            bufs[b][1] = max(pts.nbytes, 16)
                        ctx.check(lib.octl_dev_upload_async(ctx.handle, bufs[b][0], nat.ptr(pts), pts.nbytes))
                        ctx.check(lib.octl_ctx_sync_uploads(ctx.handle))   # this thread waits, nobody else does
                        cloud = _StagedCloud(ctx, bufs[b][0], len(pts), lambda b=b: self._free.put(b))
                except BaseException as e:
                    self._free.put(b)
                    fut.set_exception(e)
                    continue
                self._jobs[index % len(self._jobs)].put((job, cloud))
        finally:
            for q in self._jobs:
                q.put(None)
            # (the workers have drained their queues when close() joins them; the buffers go back then)
            self._staging_to_free = (ctx, bufs)

    def _worker(self, w):
        from octreelib_amd.grid import Grid, GridConfig

        ctx = nat.Context(self._device)
        try:
            with nat.use_context(ctx):
                while True:
                    item = self._jobs[w].get()
                    if item is None:
                        return
                    (points, fn, index, fut), cloud = item
                    grid = None
                    try:
                        grid = Grid(GridConfig(voxel_edge_length=self._edge))
                        grid.insert_points(0, cloud)
                        fut.set_result(fn(grid, index))
                    except BaseException as e:  # delivered to whoever waits for the result
                        fut.set_exception(e)
                    finally:
                        if grid is not None:
                            grid._forest.close()   # (synchronises this context: the staging buffer is free again)
                        if cloud is not points:
                            cloud.release()
        finally:
            ctx.close()

    def submit(self, points, fn):
        """Queue one scan; returns a concurrent.futures.Future of fn(grid, index).  `points` must stay unchanged
        until the future is done (it is uploaded when a staging buffer is free)."""
        from concurrent.futures import Future

        if self._closed:
            raise RuntimeError("ScanPipeline is closed")
        fut = Future()
        i = self._next
        self._next += 1
        self._inbox.put((points, fn, i, fut))
        return fut

    def map(self, scans, fn, depth=None):
        """fn(grid, index) for every scan, results in submission order.  At most `depth` scans (default: one per
        context being fitted plus one being uploaded) are queued or running: before scan k is submitted the result
        of scan k - depth has been delivered.  A front end that refills host staging arrays therefore needs a ring
        of depth + 1 of them (the iterator is asked for scan k before that wait)."""
        from collections import deque

        depth = len(self._jobs) + 1 if depth is None else max(1, int(depth))
        pending = deque()
        for pts in scans:
            if len(pending) >= depth:
                yield pending.popleft().result()
            pending.append(self.submit(pts, fn))
        while pending:
            yield pending.popleft().result()

    def close(self):
        if self._closed:
            return
        self._closed = True
        self._inbox.put(None)
        for t in self._threads:
            t.join()
        held = getattr(self, "_staging_to_free", None)
        if held is not None:
            ctx, bufs = held
            for ptr, cap in bufs:
                if ptr.value:
                    ctx.lib.octl_dev_free(ctx.handle, ptr)
            ctx.close()
            self._staging_to_free = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
