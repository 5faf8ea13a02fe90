"""Device-wide primitives (scan, stable radix sort) through their C-ABI test hooks."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from octreelib_amd import _native as nat

    return nat.get_context()


@pytest.mark.parametrize("n", [0, 1, 7, 2048, 2049, 100_000, 3_000_001])
def test_exclusive_scan(ctx, n):
    from octreelib_amd import _native as nat

    rng = np.random.default_rng(n)
    a = rng.integers(0, 1000, n, dtype=np.uint32)
    out = np.empty(n, dtype=np.uint32)
    total = C.c_uint32(0)
    ctx.check(ctx.lib.octl_debug_exclusive_scan(ctx.handle, nat.ptr(a), n, nat.ptr(out), C.byref(total)))
    want = np.concatenate(([0], np.cumsum(a, dtype=np.uint64)[:-1])).astype(np.uint32) if n else a
    assert np.array_equal(out, want)
    assert total.value == int(a.sum(dtype=np.uint64) & 0xFFFFFFFF)


@pytest.mark.parametrize("n,bits", [(1, 8), (63, 3), (2048, 8), (5000, 17), (250_000, 24), (1_000_003, 40)])
def test_radix_sort_is_stable(ctx, n, bits):
    from octreelib_amd import _native as nat

    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 1 << bits, n, dtype=np.uint64)
    if n > 100:
        keys[: n // 3] = keys[0]  # many duplicates: stability matters
    vals = np.arange(n, dtype=np.uint32)
    k2, v2 = keys.copy(), vals.copy()
    ctx.check(ctx.lib.octl_debug_radix_sort(ctx.handle, nat.ptr(k2), nat.ptr(v2), n, bits))
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k2, keys[order])
    assert np.array_equal(v2, vals[order])
