"""
Test tooling (NOT product code, NOT reference code): makes the upstream octreelib
importable in the build container so that golden vectors can be generated from it.

The upstream package fails to import here for ordinary Python reasons only:
  * NumPy >= 2 removed ``np.float_``           (upstream pins numpy ^1.26)
  * ``k3d`` (HTML plotting) is not installed    (only used by Grid.visualize)
  * ``numba`` is not installed                  (only used by the RANSAC kernel)

``install()`` therefore
  1. aliases ``np.float_ = np.float64``,
  2. registers an empty ``k3d`` module,
  3. registers a small stand-in for the part of ``numba`` / ``numba.cuda``'s *simulator*
     API that the upstream kernel uses (threadIdx/blockIdx/blockDim, local/shared arrays,
     syncthreads, atomic.max, atomic.compare_and_swap, to_device/copy_to_host,
     ``kernel[grid, block](...)``), executing every CUDA thread of a block as a Python
     thread.  The upstream kernel *source* then runs unmodified.

Nothing from the upstream tree is copied; the upstream tree is only imported from
``/root/reference`` (absent on the GPU box — tests that need it skip there).
"""

import sys
import threading
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


# --------------------------------------------------------------------------------------
# numba.cuda simulator stand-in
# --------------------------------------------------------------------------------------
class _Dim3:
    def __init__(self, x=0, y=0, z=0):
        self.x, self.y, self.z = x, y, z


class _ThreadState(threading.local):
    def __init__(self):
        self.threadIdx = _Dim3()
        self.blockIdx = _Dim3()
        self.blockDim = _Dim3()
        self.block = None
        self.shared_call = 0


_state = _ThreadState()


class _Block:
    """Per-block shared state: barrier, shared arrays (by call order), a lock for atomics."""

    def __init__(self, n_threads):
        self.barrier = threading.Barrier(n_threads)
        self.shared = {}
        self.lock = threading.Lock()


_atomic_lock = threading.Lock()

# Observables the upstream API never returns (CudaRansac.evaluate only hands back the mask,
# cuda_ransac.py:80-81): when RECORD is a list, every launched block appends the final contents
# of its shared arrays in declaration order - for the RANSAC kernel (cuda_ransac.py:125-128)
# [best_plane f32[4], max_inliers_number i32[1], mutex i32[1]]; a block that returned before
# declaring them (n < k, cuda_ransac.py:96-97) appends an empty list.
RECORD = None


class _DeviceArray(np.ndarray):
    def copy_to_host(self):
        return np.array(self)


def _to_device(arr):
    return np.array(arr).view(_DeviceArray)


class _Local:
    @staticmethod
    def array(shape, dtype):
        return np.zeros(shape, dtype=dtype)


class _Shared:
    @staticmethod
    def array(shape, dtype):
        block = _state.block
        key = _state.shared_call
        _state.shared_call += 1
        with block.lock:
            if key not in block.shared:
                block.shared[key] = np.zeros(shape, dtype=dtype)
            return block.shared[key]


class _Atomic:
    @staticmethod
    def max(arr, idx, val):
        with _atomic_lock:
            old = arr[idx]
            if val > old:
                arr[idx] = val
            return old

    @staticmethod
    def compare_and_swap(arr, old, val):
        with _atomic_lock:
            cur = arr[0]
            if cur == old:
                arr[0] = val
            return cur


def _syncthreads():
    _state.block.barrier.wait()


class _Kernel:
    def __init__(self, fn):
        self._fn = fn

    def __getitem__(self, cfg):
        grid, block = cfg
        grid = grid[0] if isinstance(grid, tuple) else int(grid)
        block = block[0] if isinstance(block, tuple) else int(block)

        def launch(*args):
            for b in range(grid):
                blk = _Block(block)
                errors = []

                def run(t, blk=blk, b=b):
                    _state.threadIdx = _Dim3(t)
                    _state.blockIdx = _Dim3(b)
                    _state.blockDim = _Dim3(block)
                    _state.block = blk
                    _state.shared_call = 0
                    try:
                        self._fn(*args)
                    except threading.BrokenBarrierError:
                        pass
                    except BaseException as e:  # noqa: BLE001 - re-raised in the launcher
                        errors.append(e)
                        blk.barrier.abort()

                threads = [threading.Thread(target=run, args=(t,)) for t in range(block)]
                for th in threads:
                    th.start()
                for th in threads:
                    th.join()
                if errors:
                    raise errors[0]
                if RECORD is not None:
                    RECORD.append([np.array(blk.shared[key]) for key in sorted(blk.shared)])

        return launch

    def __call__(self, *args, **kwargs):  # device functions are plain calls
        return self._fn(*args, **kwargs)


def _jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return _Kernel(args[0])

    def deco(fn):
        if kwargs.get("device"):
            return fn
        return _Kernel(fn)

    return deco


class _CudaModule(types.ModuleType):
    """`cuda.threadIdx` etc. must resolve per Python thread."""

    @property
    def threadIdx(self):
        return _state.threadIdx

    @property
    def blockIdx(self):
        return _state.blockIdx

    @property
    def blockDim(self):
        return _state.blockDim


def _make_numba():
    nb = types.ModuleType("numba")
    nb.int32 = np.int32
    nb.int64 = np.int64
    nb.float32 = np.float32
    nb.float64 = np.float64
    nb.size_t = np.uint64
    cuda = _CudaModule("numba.cuda")
    cuda.jit = _jit
    cuda.to_device = _to_device
    cuda.local = _Local
    cuda.shared = _Shared
    cuda.atomic = _Atomic
    cuda.syncthreads = _syncthreads
    nb.cuda = cuda
    return nb, cuda


def install(reference_root: str = REFERENCE_ROOT):
    """Make ``import octreelib`` resolve to the upstream package at ``reference_root``."""
    sys.dont_write_bytecode = True  # the reference mount is read-only
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    if "k3d" not in sys.modules:
        sys.modules["k3d"] = types.ModuleType("k3d")
    if "numba" not in sys.modules:
        nb, cuda = _make_numba()
        sys.modules["numba"] = nb
        sys.modules["numba.cuda"] = cuda
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


def reference_available(reference_root: str = REFERENCE_ROOT) -> bool:
    import os

    return os.path.isdir(os.path.join(reference_root, "octreelib"))
