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


def _plane_arith(num3, den, c, k, certified=False):
    from octreelib_amd import _native as nat

    ctx = nat.get_context()
    n = len(den)
    q3 = np.empty((n, 3))
    ck = np.empty(n)
    sq = np.empty(n)
    fn = ctx.lib.octl_debug_plane_arith_certified if certified else ctx.lib.octl_debug_plane_arith
    ctx.check(fn(ctx.handle, nat.ptr(np.ascontiguousarray(num3)), nat.ptr(np.ascontiguousarray(den)),
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


def _certified_cases(seed, n=2_000_000):
    """Operands inside the ranges the block certificate of csrc/ransac.hip (coord_in_fast_range) guarantees:
    norm in [2^-200, 2^138], numerators zero or in [2^-552, norm (1 + 2^-52)], centroid sums +0.0 or in
    [2^-82, 2^35), s in [2^-400, 2^276] - with the edges of every range over-represented."""
    rng = np.random.default_rng(seed)
    mant = rng.random(n) + 1.0
    pick = rng.random(n)
    mant = np.where(pick < 0.05, np.nextafter(2.0, 0.0), mant)
    mant = np.where((pick >= 0.05) & (pick < 0.10), 1.0, mant)
    mant = np.where((pick >= 0.10) & (pick < 0.15), 1.0 + 2.0 ** -52, mant)
    e_den = rng.integers(-200, 138, n)
    e_den = np.where(rng.random(n) < 0.1, rng.choice([-200, -199, 136, 137], n), e_den)
    den = np.ldexp(mant, e_den)
    den = np.minimum(den, 2.0 ** 138)
    num3 = (rng.random((n, 3)) * 2 - 1) * den[:, None]
    # tiny numerators down to the grid of the cofactors (2^-552), exact zeros of both signs, |a| = norm
    tiny = np.ldexp(rng.random((n, 3)) + 1.0, rng.integers(-552, -100, (n, 3))) * rng.choice([-1.0, 1.0], (n, 3))
    tiny = np.where(np.abs(tiny) <= den[:, None], tiny, den[:, None])
    sp = rng.random((n, 3))
    num3 = np.where(sp < 0.25, tiny, num3)
    num3 = np.where((sp >= 0.25) & (sp < 0.30), 0.0, num3)
    num3 = np.where((sp >= 0.30) & (sp < 0.33), -0.0, num3)
    num3 = np.where((sp >= 0.33) & (sp < 0.36), den[:, None] * np.array([1.0, -1.0, 1.0]), num3)
    num3 = np.where((sp >= 0.36) & (sp < 0.38), np.nextafter(den, np.inf)[:, None], num3)
    num3 = np.where((np.abs(num3) < 2.0 ** -552) & (num3 != 0), np.copysign(2.0 ** -552, num3), num3)
    # centroid sums: multiples of 2^-82 below 2^35 (what sums of certified coordinates are), incl. +0.0
    c = np.ldexp(rng.random(n) + 1.0, rng.integers(-82, 34, n)) * rng.choice([-1.0, 1.0], n)
    c = np.where(rng.random(n) < 0.3, (rng.random(n) * 2 - 1) * 64.0, c)
    c = np.where(np.abs(c) < 2.0 ** -82, 2.0 ** -82, c)
    c = np.where(rng.random(n) < 0.03, 0.0, c)
    # the square root's operands
    s = np.ldexp(rng.random(n) + 1.0, rng.integers(-400, 275, n))
    s = np.where(rng.random(n) < 0.05, rng.choice([2.0 ** -400, 2.0 ** 276, 1.0, 4.0, np.nextafter(4.0, 0)], n), s)
    return num3, den, c, s


@pytest.mark.parametrize("seed", [10, 11, 12])
def test_certified_plane_fit_arithmetic_is_ieee_without_guards(seed):
    """A block whose coordinates are all +0.0 or in [2^-30, 2^31) runs the plane fit's division / square
    root shortcuts WITHOUT their per-lane range guards (csrc/ransac.hip, coord_in_fast_range).  Over the operand
    ranges that certificate implies the unguarded forms must still be the IEEE results."""
    num3, den, c, s = _certified_cases(seed)
    for k in (1, 3, 6, 7, 16):
        q3, ck, _ = _plane_arith(num3, den, c, k, certified=True)
        assert _same_bits(q3, num3 / den[:, None]).all()
        assert _same_bits(ck, c / np.float64(k)).all()
    _, _, sq = _plane_arith(num3, den, s, 6, certified=True)
    assert _same_bits(sq, np.sqrt(s)).all()


def test_blocks_outside_the_range_certificate_keep_the_guards():
    """-0.0, denormal-scale and astronomically large coordinates make a block fail the certificate: the
    guarded fit must still reproduce the oracle bit for bit."""
    from octreelib_amd.ransac import CudaRansac
    from oracle import ransac_np as rnp

    rng = np.random.default_rng(5)
    blocks = []
    base = rng.random((40, 3))
    base[:, 2] = 0.3 * base[:, 0] + 0.1 * base[:, 1] + rng.normal(0, 0.003, 40)
    blocks.append(base.copy())                                  # certified
    b = base.copy(); b[3, 0] = -0.0; b[7, 1] = -0.0; blocks.append(b)          # a negative zero
    b = base.copy() * 2.0 ** -40; blocks.append(b)              # below 2^-30
    b = base.copy() * 2.0 ** 40; blocks.append(b)               # above 2^31
    b = base.copy(); b[:, 0] = 0.0; blocks.append(b)            # a plane of exact +0.0: certified, zero sums
    b = base.copy(); b[:, 0] = -0.0; blocks.append(b)           # ... of -0.0: not certified
    b = base.copy(); b[5] = 1e-310; blocks.append(b)            # a denormal point
    b = np.tile(base[:1], (12, 1)); blocks.append(b)            # identical points: norm == 0 (util.py:77-78)
    sizes = np.array([len(b) for b in blocks], dtype=np.int32)
    cloud = np.vstack(blocks)
    for thr in (0.01, 0.01 * 2.0 ** -40, 0.01 * 2.0 ** 40):
        np.random.seed(7)
        op = CudaRansac(threshold=thr, hypotheses_number=1024, initial_points_number=6)
        mask, planes, counts, index = op.evaluate(cloud, sizes, details=True)
        o_mask, o_count, o_plane, o_index, _ = rnp.evaluate(cloud, sizes, op.random_hypotheses, thr, details=True)
        assert np.array_equal(counts, o_count)
        assert np.array_equal(index, o_index)
        assert np.array_equal(planes.view(np.uint32), o_plane.view(np.uint32))
        assert np.array_equal(mask, o_mask)


def test_context_switches_counters_and_device_identity():
    """Round-5 additions to the C ABI: diagnostic switches of a live context (octl_debug_set_option - the library
    reads its OCTL_* environment only when a context is created), the launch counter beside the host-wait counter,
    the identity of the device behind a context, and the communicator query without a communicator."""
    import ctypes as C

    from octreelib_amd import _native as nat
    from octreelib_amd import synthetic
    from octreelib_amd._engine import Forest

    ctx = nat.get_context()
    lib = ctx.lib
    with pytest.raises(ValueError, match="no such option"):
        ctx.set_option("NO_SUCH_SWITCH", 1)
    ctx.set_option("OCTL_NO_SPIN_WAIT", 1)      # (the prefix is accepted)
    ctx.set_option("NO_SPIN_WAIT", 0)
    l0, l1, s0, s1 = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    ctx.check(lib.octl_debug_launches(C.byref(l0)))
    ctx.check(lib.octl_debug_host_syncs(C.byref(s0)))
    f = Forest(0, np.zeros(3), 1.0)
    f.add_pose(synthetic.planar_cloud(20_000, (3, 3, 3), seed=1, stream=9))
    f.subdivide(32)
    f.close()
    ctx.check(lib.octl_debug_launches(C.byref(l1)))
    ctx.check(lib.octl_debug_host_syncs(C.byref(s1)))
    # (how many exactly depends on what the process-wide context has built before - hints, sparse-scene and chunk
    #  history: the counters must move, by a few dozen at most)
    assert 5 <= l1.value - l0.value <= 80 and 1 <= s1.value - s0.value <= 30
    bus = C.create_string_buffer(32)
    uu = (C.c_uint8 * 16)()
    cus = C.c_int32(0)
    ctx.check(lib.octl_device_identity(ctx.handle, bus, C.cast(uu, C.c_void_p), C.byref(cus)))
    assert len(bus.value.decode().split(":")) == 3 and cus.value >= 64 and any(bytes(uu))
    n = C.c_int32(0)
    assert lib.octl_comm_info(ctx.handle, C.byref(n), None, None) == nat.OCTL_E_STATE   # no communicator on this context
