"""Device-wide primitives (scan, stable radix sort) through their C-ABI test hooks."""

import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from octreelib_amd import _native as nat

    return nat.get_context()


@pytest.mark.parametrize("n", [0, 1, 7, 2048, 2049, 100_000, 2_097_152, 2_097_153, 3_000_001])
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


def test_exclusive_scan_back_to_back(ctx):
    """The single-pass scan reuses its per-tile status words across calls (an epoch tag makes the
    words of earlier scans read as unpublished): many scans of changing sizes in a row."""
    from octreelib_amd import _native as nat

    rng = np.random.default_rng(99)
    for n in [500_000, 4096, 1_999_999, 2048, 70_000, 1_200_000, 3, 650_000] * 3:
        a = rng.integers(0, 5000, n, dtype=np.uint32)
        out = np.empty(n, dtype=np.uint32)
        total = C.c_uint32(0)
        ctx.check(ctx.lib.octl_debug_exclusive_scan(ctx.handle, nat.ptr(a), n, nat.ptr(out), C.byref(total)))
        cs = np.cumsum(a, dtype=np.uint64)
        want = (np.concatenate((np.zeros(1, dtype=np.uint64), cs[:-1])) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        assert np.array_equal(out, want)  # sums wrap modulo 2^32
        assert total.value == int(cs[-1] & np.uint64(0xFFFFFFFF))


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


def _same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    both_nan = np.isnan(a) & np.isnan(b)
    return (a.view(np.uint64) == b.view(np.uint64)) | both_nan


def _plane_arith(num3, den, c, k):
    from octreelib_amd import _native as nat

    ctx = nat.get_context()
    n = len(den)
    q3 = np.empty((n, 3))
    ck = np.empty(n)
    sq = np.empty(n)
    ctx.check(ctx.lib.octl_debug_plane_arith(ctx.handle, nat.ptr(np.ascontiguousarray(num3)), nat.ptr(np.ascontiguousarray(den)),
                                             nat.ptr(np.ascontiguousarray(c)), k, n, nat.ptr(q3), nat.ptr(ck), nat.ptr(sq)))
    return q3, ck, sq


def _division_cases(seed, n=2_000_000):
    rng = np.random.default_rng(seed)
    mant = rng.random(n) + 1.0
    pick = rng.random(n)
    mant = np.where(pick < 0.05, np.nextafter(2.0, 0.0), mant)   # significand all ones
    mant = np.where((pick >= 0.05) & (pick < 0.10), 1.0, mant)
    mant = np.where((pick >= 0.10) & (pick < 0.15), 1.0 + 2.0 ** -52, mant)
    den = np.ldexp(mant, rng.integers(-320, 320, n))
    den = np.where(rng.random(n) < 0.02, np.ldexp(mant, rng.integers(-1070, 1020, n)), den)
    den[den == 0] = 1.0
    num3 = (rng.random((n, 3)) * 2 - 1) * den[:, None]              # |a| <= norm as in the plane fit
    scale = np.ldexp(1.0, rng.integers(-700, 40, (n, 3)))
    with np.errstate(all="ignore"):
        num3 = np.where(rng.random((n, 3)) < 0.3, num3 * scale, num3)   # tiny and (rarely) larger numerators
    # (an infinite numerator over a finite norm cannot occur in the plane fit - norm >= max |a_i| -
    #  and is outside the shortcut's domain; with an infinite norm everything takes the true division)
    special = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-310, np.nan, 1.0, -1.0])
    special_c = np.array([0.0, -0.0, 5e-324, -5e-324, 1e-310, np.inf, -np.inf, np.nan, 1.0, -1.0])
    sp = rng.random((n, 3))
    num3 = np.where(sp < 0.05, special[rng.integers(0, len(special), (n, 3))], num3)
    num3 = np.where((sp >= 0.05) & (sp < 0.08), den[:, None] * np.array([1.0, -1.0, 1.0]), num3)
    num3[np.isinf(num3)] = 1.0
    den[:1000] = np.inf
    num3[:500] = np.array([np.inf, -np.inf, 3.0])
    k = int(rng.integers(1, 17))
    c = np.ldexp(rng.random(n) * 2 - 1, rng.integers(-1074, 1023, n))
    c = np.where(rng.random(n) < 0.5, (rng.random(n) * 2 - 1) * 64.0, c)
    c = np.where(rng.random(n) < 0.02, special_c[rng.integers(0, len(special_c), n)], c)
    return num3, den, c, k


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_plane_fit_division_shortcuts_are_ieee_divisions(seed):
    """csrc/ransac.hip divides the plane normal by its norm with one shared reciprocal, the centroid
    by k with a two-term reciprocal and takes the norm's square root without the library's range
    handling: all must be the correctly rounded results the reference computes
    (util.py:42-44,76,80-82), for every input incl. zeros, denormals, inf, NaN."""
    num3, den, c, k = _division_cases(seed)
    q3, ck, sq = _plane_arith(num3, den, c, k)
    with np.errstate(all="ignore"):
        assert _same_bits(q3, num3 / den[:, None]).all()
        assert _same_bits(ck, c / np.float64(k)).all()
        assert _same_bits(sq, np.sqrt(c)).all()


def test_plane_fit_divisions_near_rounding_midpoints():
    """Quotients engineered to sit next to a midpoint between two doubles (the hard cases for
    any reciprocal-based division)."""
    import random

    random.seed(3)
    n = 60_000
    num = np.empty((n, 3))
    den = np.empty(n)
    M = 1 << 54
    for i in range(n):
        b = random.getrandbits(52) | (1 << 52) | 1          # odd 53-bit divisor
        binv = pow(b, -1, M)
        e = random.randint(-60, 60)
        den[i] = float(b) * 2.0 ** e
        for j in range(3):
            while True:
                r = random.choice((1, -1, 3, -3, 5, -7))
                m = (-r * binv) % M                         # m*b + r == 0 (mod 2^54)
                if m >= (1 << 53) and (m & 1):
                    break
            a = (m * b + r) >> 54                           # exact; a/b = (m + r/b) / 2^54, m odd, 54 bits:
            assert a.bit_length() <= 53                     # a rounding midpoint +- 2^-107 relative
            num[i, j] = float(a) * 2.0 ** e * random.choice((1.0, -1.0, 0.5, 0.25))
    c = np.array([float(random.getrandbits(53)) for _ in range(n)])
    for k in (3, 5, 6, 7, 9, 11, 13):
        q3, ck, sq = _plane_arith(num, den, c, k)
        assert _same_bits(q3, num / den[:, None]).all()
        assert _same_bits(ck, c / np.float64(k)).all()
        assert _same_bits(sq, np.sqrt(c)).all()
