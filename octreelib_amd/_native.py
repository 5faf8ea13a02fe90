"""
ctypes binding of liboctree_hip.so (include/octreelib_hip.h).  No PyTorch, no CPU fallback:
if the HIP library is missing or no GPU is visible the product path fails loudly.
"""

import ctypes as C
import os
import threading

import numpy as np

_LIB_PATH = os.environ.get(
    "OCTREELIB_AMD_LIB",  # kernel-variant experiments only
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liboctree_hip.so"),
)

OCTL_OK = 0
OCTL_E_INVALID = -1
OCTL_E_HIP = -2
OCTL_E_NOMEM = -3
OCTL_E_DOMAIN = -4
OCTL_E_DEPTH = -5
OCTL_E_STATE = -6
OCTL_E_COMM = -7
UNIQUE_ID_BYTES = 128


class NativeLibraryError(RuntimeError):
    """liboctree_hip.so is missing / not loadable, or no HIP device is visible."""


class DomainError(ValueError, IndexError):
    """Input outside the parity domain (the reference raises IndexError or silently picks a
    wrong child for a point outside the cube of a node that is being split)."""


class BuildInfo(C.Structure):
    _fields_ = [
        ("n_points", C.c_int64),
        ("n_voxels", C.c_int64),
        ("n_nodes", C.c_int64),
        ("n_internal", C.c_int64),
        ("n_blocks", C.c_int64),
        ("max_depth", C.c_int32),
        ("n_levels", C.c_int32),
    ]


_p = C.c_void_p
_i32, _i64, _f64 = C.c_int32, C.c_int64, C.c_double
_pi32, _pi64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

# name -> (restype, argtypes): every symbol include/octreelib_hip.h declares
SIGNATURES = {
    "octl_abi_version": (C.c_int, []),
    "octl_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "octl_ctx_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "octl_ctx_destroy": (None, [_p]),
    "octl_last_error": (C.c_char_p, [_p]),
    "octl_ctx_sync": (C.c_int, [_p]),
    "octl_ctx_set_profiling": (C.c_int, [_p, C.c_int]),
    "octl_ctx_get_timings": (C.c_int, [_p, _p, C.c_int, _p, _p, C.c_int, C.POINTER(C.c_int)]),
    "octl_forest_create": (C.c_int, [_p, C.c_int, _p, _f64, C.POINTER(_p)]),
    "octl_forest_destroy": (None, [_p]),
    "octl_forest_clear": (C.c_int, [_p]),
    "octl_forest_add_pose": (C.c_int, [_p, _p, _i64, _pi32]),
    "octl_forest_add_pose_device": (C.c_int, [_p, _p, _i64, _pi32]),
    "octl_forest_add_pose_adopt": (C.c_int, [_p, _p, _i64, _pi32]),
    "octl_forest_set_contents": (C.c_int, [_p, _i64, _p, _p, _p, _p]),
    "octl_forest_extend_pose": (C.c_int, [_p, _i32, _p, _i64]),
    "octl_forest_extend_pose_device": (C.c_int, [_p, _i32, _p, _i64]),
    "octl_forest_build": (C.c_int, [_p, _i64, _p, _i32, _i32, _i32, C.POINTER(BuildInfo)]),
    "octl_forest_set_scheme": (C.c_int, [_p, _p, _p, _i64, _i32]),
    "octl_forest_get_nodes": (C.c_int, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _pi64]),
    "octl_forest_get_voxels": (C.c_int, [_p, _i64, _p, _pi64]),
    "octl_forest_get_blocks": (C.c_int, [_p, _i64, _p, _p, _p, _p, _pi64]),
    "octl_forest_get_slot_voxels": (C.c_int, [_p, _i32, _i64, _p, _pi64]),
    "octl_forest_slot_counts": (C.c_int, [_p, _i32, _pi64, _pi64]),
    "octl_forest_internal_per_voxel": (C.c_int, [_p, _i64, _p, _pi64]),
    "octl_forest_get_perm": (C.c_int, [_p, _i64, _p, _pi64]),
    "octl_forest_get_points": (C.c_int, [_p, _i64, _i64, _p]),
    "octl_forest_gather_blocks": (C.c_int, [_p, _p, _i64, _i64, _p, _pi64]),
    "octl_forest_ransac": (C.c_int, [_p, _p, _i64, _p, _i32, _i32, _f64, _p, _p, _p]),
    "octl_forest_reference_order": (C.c_int, [_p, _p, _i32, _i64, _p, _pi64]),
    "octl_forest_ransac_all": (C.c_int, [_p, _i32, _p, _i32, _p, _i32, _i32, _f64]),
    "octl_forest_get_mask": (C.c_int, [_p, _i64, _p, _pi64]),
    "octl_forest_apply_mask": (C.c_int, [_p, _pi64]),
    "octl_forest_apply_mask_async": (C.c_int, [_p]),
    "octl_forest_settle": (C.c_int, [_p, _pi64]),
    "octl_forest_apply_host_mask": (C.c_int, [_p, _p, _i64, _pi64]),
    "octl_forest_filter_count": (C.c_int, [_p, _p, _i32, _i64, _i64, _pi64]),
    "octl_ransac_evaluate": (
        C.c_int,
        [_p, _p, _i64, _p, _i64, _p, _i32, _i32, _f64, _p, _p, _p, _p],
    ),
    "octl_voxel_owner": (_i32, [_i64, _i64, _i64, _i32]),
    "octl_comm_unique_id": (C.c_int, [_p]),
    "octl_comm_init": (C.c_int, [_p, _i32, _i32, _p]),
    "octl_comm_destroy": (C.c_int, [_p]),
    "octl_comm_info": (C.c_int, [_p, _pi32, _pi32, _pi32]),
    "octl_device_identity": (C.c_int, [_p, C.c_char_p, _p, _pi32]),
    "octl_route_points": (C.c_int, [_p, _p, _p, _i64, _i64, _p, _f64, _pi64, _p]),
    "octl_forest_add_pose_routed": (C.c_int, [_p, _pi32]),
    "octl_forest_add_pose_routed_from": (C.c_int, [_p, _p, _pi32]),
    "octl_debug_route_partition": (C.c_int, [_p, _p, _i64, _i64, C.c_double, C.c_int32, _p, _p, _p]),
    "octl_debug_host_syncs": (C.c_int, [C.POINTER(C.c_uint64)]),
    "octl_debug_launches": (C.c_int, [C.POINTER(C.c_uint64)]),
    "octl_debug_spec_finish": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "octl_debug_fail_alloc": (C.c_int, [_i64, _pi64]),
    "octl_debug_set_option": (C.c_int, [_p, C.c_char_p, _i64]),
    "octl_debug_key_geometry": (C.c_int, [_p, C.c_uint64, _i64, C.c_uint32, _i32, _p, _p, _p, _p, _p]),
    "octl_route_get_gidx": (C.c_int, [_p, _i64, _p, _pi64]),
    "octl_comm_allreduce_i64": (C.c_int, [_p, _p, _i32]),
    "octl_dev_alloc": (C.c_int, [_p, _i64, C.POINTER(_p)]),
    "octl_dev_free": (C.c_int, [_p, _p]),
    "octl_dev_upload": (C.c_int, [_p, _p, _p, _i64]),
    "octl_dev_download": (C.c_int, [_p, _p, _p, _i64]),
    "octl_host_alloc": (C.c_int, [_p, _i64, C.POINTER(_p)]),
    "octl_host_free": (C.c_int, [_p, _p]),
    "octl_dev_upload_async": (C.c_int, [_p, _p, _p, _i64]),
    "octl_ctx_sync_uploads": (C.c_int, [_p]),
    "octl_dev_copy_bandwidth": (C.c_int, [_p, _i64, C.c_int, C.POINTER(_f64)]),
    "octl_debug_exclusive_scan": (C.c_int, [_p, _p, _i64, _p, _p]),
    "octl_debug_radix_sort": (C.c_int, [_p, _p, _p, _i64, C.c_int]),
    "octl_debug_plane_arith": (C.c_int, [_p, _p, _p, _p, C.c_int32, _i64, _p, _p, _p]),
    "octl_debug_plane_arith_certified": (C.c_int, [_p, _p, _p, _p, C.c_int32, _i64, _p, _p, _p]),
}

_lib = None
_lib_lock = threading.Lock()


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load liboctree_hip.so and bind every declared symbol.  Loading does not touch the GPU."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise NativeLibraryError(
                f"{_LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
                "octreelib_amd has no CPU fallback."
            )
        try:
            lib = C.CDLL(_LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise NativeLibraryError(f"cannot load {_LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def ptr(a):
    """void* of a C-contiguous NumPy array (None -> NULL)."""
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One device context (device + HIP stream + scratch).  Not thread safe."""

    def __init__(self, device: int = 0):
        self.lib = load()
        n = C.c_int(0)
        self.lib.octl_device_count(C.byref(n))
        if n.value <= 0:
            raise NativeLibraryError(
                "no HIP device is visible: octreelib_amd runs on an MI355X only (no CPU fallback)"
            )
        h = C.c_void_p()
        rc = self.lib.octl_ctx_create(int(device), C.byref(h))
        if rc != OCTL_OK or not h.value:
            raise NativeLibraryError(f"octl_ctx_create(device={device}) failed with code {rc}")
        self.handle = h
        self.device = int(device)

    def check(self, rc: int):
        if rc == OCTL_OK:
            return
        msg = self.lib.octl_last_error(self.handle)
        msg = msg.decode("utf-8", "replace") if msg else f"error {rc}"
        if rc == OCTL_E_INVALID:
            raise ValueError(msg)
        if rc == OCTL_E_NOMEM:
            raise MemoryError(msg)
        if rc == OCTL_E_DOMAIN:
            raise DomainError(msg)
        if rc == OCTL_E_DEPTH:
            # the reference dies with RecursionError on duplicate points + count criterion
            raise RecursionError(msg)
        raise RuntimeError(msg)

    def sync(self):
        self.check(self.lib.octl_ctx_sync(self.handle))

    def set_option(self, name: str, value=1):
        """A diagnostic switch of this context (include/octreelib_hip.h: octl_debug_set_option), e.g.
        set_option("NO_BUCKET_BUILD", 1).  The library reads OCTL_<NAME> from the environment only when a context
        is created; tests and A/B runs that compare code paths on a live context go through here.  None / False = 0."""
        v = 0 if value is None or value is False else int(value)
        self.check(self.lib.octl_debug_set_option(self.handle, name.encode(), v))
        touched = self.__dict__.setdefault("_options_touched", set())
        touched.add(name)

    def reset_options(self):
        for name in list(self.__dict__.get("_options_touched", ())):
            self.lib.octl_debug_set_option(self.handle, name.encode(), 0)
        self.__dict__["_options_touched"] = set()

    def set_profiling(self, enabled: bool):
        # (True / 1: every timed region; 2: the RANSAC kernel only; False / 0: off)
        self.check(self.lib.octl_ctx_set_profiling(self.handle, int(enabled)))

    def timings(self):
        """{kernel name: (total ms, launches)} since profiling was enabled."""
        cap, stride = 64, 48
        names = C.create_string_buffer(cap * stride)
        ms = (C.c_float * cap)()
        launches = (C.c_int64 * cap)()
        n = C.c_int(0)
        self.check(
            self.lib.octl_ctx_get_timings(
                self.handle, C.cast(names, C.c_void_p), stride, C.cast(ms, C.c_void_p),
                C.cast(launches, C.c_void_p), cap, C.byref(n),
            )
        )
        out = {}
        for i in range(min(n.value, cap)):
            name = names.raw[i * stride : (i + 1) * stride].split(b"\0", 1)[0].decode()
            out[name] = (float(ms[i]), int(launches[i]))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.octl_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_device() -> int:
    for var in ("OCTREELIB_AMD_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v.strip().lstrip("-").isdigit():
            return int(v)
    return 0


_thread_ctx = threading.local()


class use_context:
    """`with use_context(ctx):` - every Grid / Octree / OctreeManager / CudaRansac / DeviceCloud created by THIS
    thread inside the block lives on `ctx` instead of the process-wide context (a context is one HIP stream and is
    not thread safe: a second host thread that wants to overlap its scans with the first one's brings its own,
    octreelib_amd.feed.ScanPipeline)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def __enter__(self):
        self._prev = getattr(_thread_ctx, "ctx", None)
        _thread_ctx.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _thread_ctx.ctx = self._prev


def get_context(device=None) -> Context:
    """The calling thread's context (use_context) or the process-wide context of a device (created on first use)."""
    if device is None:
        ctx = getattr(_thread_ctx, "ctx", None)
        if ctx is not None:
            return ctx
    device = default_device() if device is None else int(device)
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx


def as_points(points) -> np.ndarray:
    """(n,3) C-contiguous float64 — the reference upcasts every input to f64
    (internal/voxel.py:81-83, octree.py:100)."""
    a = np.ascontiguousarray(np.asarray(points, dtype=np.float64))
    if a.size == 0:
        return a.reshape(0, 3)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError(f"expected an (n, 3) point cloud, got shape {a.shape}")
    return a
