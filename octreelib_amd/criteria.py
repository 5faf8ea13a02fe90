"""
Subdivision criteria.

The reference takes arbitrary Python callables ``points -> bool`` (octree/octree.py:26); a
callable cannot run inside a HIP kernel.  The device path understands the one criterion the
library is used with - the point count, ``len(points) > K`` - given either as ``MaxPoints(K)``
or as a plain lambda/def of exactly that shape, which is recognised from its bytecode and
double-checked by probing.  Any other callable is evaluated by the host, level by level, on the
node point arrays the device produces (the predicate is the caller's Python code and can only
run there); the bucketing / partition / ordering stays on the device.
"""

import dis
import warnings
from typing import Callable, Optional, Sequence

import numpy as np

__all__ = ["MaxPoints", "count_threshold", "try_count_threshold", "try_count_interval",
           "UnsupportedCriterion"]


class UnsupportedCriterion(NotImplementedError):
    pass


class MaxPoints:
    """Subdivide a node while it holds more than `k` points: ``len(points) > k``."""

    def __init__(self, k: int):
        self.k = int(k)

    def __call__(self, points) -> bool:
        return len(points) > self.k

    def __repr__(self):
        return f"MaxPoints({self.k})"


def _resolve(fn, ins):
    """Value loaded by a LOAD_CONST / LOAD_DEREF / LOAD_GLOBAL / LOAD_FAST-less instruction."""
    if ins.opname == "LOAD_CONST":
        return ins.argval
    if ins.opname in ("LOAD_DEREF", "LOAD_CLOSURE"):
        names = fn.__code__.co_cellvars + fn.__code__.co_freevars
        idx = names.index(ins.argval) - len(fn.__code__.co_cellvars)
        return fn.__closure__[idx].cell_contents
    if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME"):
        return fn.__globals__[ins.argval]
    raise KeyError(ins.opname)


def _match_len_compare(fn) -> Optional[int]:
    """K if `fn` is `lambda p: len(p) > c` / `len(p) >= c` / `c < len(p)` / `c <= len(p)`."""
    code = getattr(fn, "__code__", None)
    if code is None or code.co_argcount != 1:
        return None
    skip = {"RESUME", "PRECALL", "PUSH_NULL", "CACHE", "COPY_FREE_VARS", "NOP"}
    ins = [i for i in dis.get_instructions(fn) if i.opname not in skip]
    arg = code.co_varnames[0]

    def is_len_call(seq):
        return (
            len(seq) == 3
            and seq[0].opname == "LOAD_GLOBAL"
            and seq[0].argval == "len"
            and seq[1].opname == "LOAD_FAST"
            and seq[1].argval == arg
            and seq[2].opname in ("CALL_FUNCTION", "CALL")
        )

    if len(ins) != 6 or ins[-1].opname != "RETURN_VALUE" or ins[-2].opname != "COMPARE_OP":
        return None
    op = ins[-2].argval
    try:
        if is_len_call(ins[0:3]):
            c = _resolve(fn, ins[3])
            if op == ">":
                return int(c) if float(c) == int(c) else int(np.floor(c))
            if op == ">=":
                return int(np.ceil(c)) - 1
        elif is_len_call(ins[1:4]):
            c = _resolve(fn, ins[0])
            if op == "<":
                return int(c) if float(c) == int(c) else int(np.floor(c))
            if op == "<=":
                return int(np.ceil(c)) - 1
    except Exception:
        return None
    return None


_FLIP = {"<": ">", "<=": ">=", ">": "<", ">=": "<=", "==": "==", "!=": "!="}
_INT_MAX = (1 << 63) - 1
_PROBE_MAX = 1 << 40  # (probe clouds are zero-stride views: any length costs 24 bytes)


def _probe_cloud(n: int) -> np.ndarray:
    """An (n, 3) f64 cloud of zeros WITHOUT its memory: a read-only zero-stride view, so that a criterion can be
    probed on both sides of a bound of any size (len() and .shape are those of a real cloud)."""
    return np.broadcast_to(np.zeros((1, 3)), (int(n), 3))


def _match_len_interval(fn):
    """(lo, hi) if `fn` is `len(p) OP c` or `c OP len(p)` with OP in < <= > >= ==: the predicate holds
    exactly when lo <= len(p) <= hi.  None otherwise."""
    if isinstance(fn, MaxPoints):
        return fn.k + 1, _INT_MAX
    code = getattr(fn, "__code__", None)
    if code is None or code.co_argcount != 1:
        return None
    skip = {"RESUME", "PRECALL", "PUSH_NULL", "CACHE", "COPY_FREE_VARS", "NOP"}
    ins = [i for i in dis.get_instructions(fn) if i.opname not in skip]
    arg = code.co_varnames[0]

    def is_len_call(seq):
        return (len(seq) == 3 and seq[0].opname == "LOAD_GLOBAL" and seq[0].argval == "len"
                and seq[1].opname == "LOAD_FAST" and seq[1].argval == arg
                and seq[2].opname in ("CALL_FUNCTION", "CALL"))

    if len(ins) != 6 or ins[-1].opname != "RETURN_VALUE" or ins[-2].opname != "COMPARE_OP":
        return None
    op = ins[-2].argval
    try:
        if is_len_call(ins[0:3]):
            c = _resolve(fn, ins[3])
        elif is_len_call(ins[1:4]):
            c = _resolve(fn, ins[0])
            op = _FLIP.get(op)
        else:
            return None
        c = float(c)
    except Exception:
        return None
    if not np.isfinite(c) or abs(c) >= 2.0 ** 62:
        return None  # len(p) < inf, len(p) > nan ...: left to the host path
    if op == ">":
        return int(np.floor(c)) + 1, _INT_MAX
    if op == ">=":
        return int(np.ceil(c)), _INT_MAX
    if op == "<":
        return 0, int(np.ceil(c)) - 1
    if op == "<=":
        return 0, int(np.floor(c))
    if op == "==" and c == int(c):
        return int(c), int(c)
    return None


def try_count_interval(criteria: Sequence[Callable]):
    """(lo, hi) such that all(criterion(points)) == (lo <= len(points) <= hi) when every criterion is a
    point-count comparison (they then run on the device, octl_forest_filter_count); None otherwise."""
    lo, hi = 0, _INT_MAX
    for c in criteria:
        iv = _match_len_interval(c)
        if iv is None:
            return None
        # double-check by probing around the bounds
        try:
            # both ends of the interval (an end beyond any allocatable cloud is taken as open)
            probes = {max(iv[0] - 1, 0), iv[0], iv[0] + 1}
            if iv[1] < _PROBE_MAX:
                probes |= {max(iv[1] - 1, 0), iv[1], iv[1] + 1}
            for n in probes:
                if n < 0 or n > _PROBE_MAX:
                    continue
                if bool(c(_probe_cloud(n))) != (iv[0] <= n <= iv[1]):
                    return None
        except Exception:
            return None
        lo, hi = max(lo, iv[0]), min(hi, iv[1])
    return lo, hi


def _probe(fn, k: int) -> bool:
    """fn must behave like len(points) > k around k."""
    try:
        for n in {0, max(k, 0), max(k, 0) + 1, max(k, 0) + 2}:
            if bool(fn(_probe_cloud(n))) != (n > k):
                return False
    except Exception:
        return False
    return True


_warned = False


def _mentions_len(fn) -> bool:
    code = getattr(fn, "__code__", None)
    return code is not None and "len" in code.co_names


def try_count_threshold(criteria: Sequence[Callable]):
    """K for a pure point-count criterion (runs entirely on the device), or None when the
    criteria are arbitrary callables (the host then evaluates them level by level).  A criterion that
    mentions len() but is not recognised gets ONE warning per process: it takes the slow host path
    (e.g. a CPython whose bytecode this matcher does not know) - MaxPoints(K) is the explicit form."""
    global _warned
    try:
        return count_threshold(criteria)
    except UnsupportedCriterion:
        if not _warned and any(_mentions_len(c) for c in criteria):
            _warned = True
            warnings.warn("octreelib_amd: a subdivision criterion that uses len(points) was not recognised as "
                          "`len(points) > K`; it is evaluated on the host level by level. Use "
                          "octreelib_amd.MaxPoints(K) for the device path.", RuntimeWarning, stacklevel=3)
        return None


_match_cache = {}


def _cached_threshold(c):
    """(k or None) for a callable, cached per code object + closure / global constant it compares with
    (the probe calls the user's function on zero arrays: once per distinct criterion, not per call)."""
    code = getattr(c, "__code__", None)
    if code is None:
        k = _match_len_compare(c)
        return k if k is not None and _probe(c, k) else None
    k = _match_len_compare(c)
    key = (code, k)
    if key not in _match_cache:
        if len(_match_cache) > 4096:
            _match_cache.clear()
        _match_cache[key] = k if (k is not None and k <= (1 << 20) and _probe(c, k)) else (
            k if (k is not None and k > (1 << 20)) else None)
    return _match_cache[key]


def count_threshold(criteria: Sequence[Callable]) -> int:
    """K such that any(criterion(points)) == len(points) > K, or raise UnsupportedCriterion.
    An empty list never subdivides (any([]) is False): K = -1 means 'never'."""
    if criteria is None:
        raise TypeError("subdivision_criteria must be a list of callables")
    ks = []
    for c in criteria:
        if isinstance(c, MaxPoints):
            ks.append(c.k)
            continue
        k = _cached_threshold(c)
        if k is None:
            raise UnsupportedCriterion(
                "only point-count criteria (octreelib_amd.MaxPoints(K) or `lambda points: "
                "len(points) > K`) run on the device; arbitrary Python callables cannot be "
                "evaluated inside a HIP kernel and there is no CPU fallback"
            )
        ks.append(k)
    if not ks:
        return -1
    if min(ks) < 0:
        # len(points) > K with K < 0 is true for empty nodes: the reference recurses forever
        raise RecursionError("the criterion is true for empty nodes: subdivision never terminates")
    return min(ks)  # any(): the smallest threshold decides
