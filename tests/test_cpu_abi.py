"""CPU-only checks of the boundary: the C-ABI library loads, exports every symbol the header
declares, and the product path refuses to run without a GPU (no CPU fallback)."""

import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "octreelib_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(octl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from octreelib_amd import _native as nat

    lib = nat.load()
    names = declared_symbols()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in octreelib_hip.h but not exported"
        assert name in nat.SIGNATURES, f"{name} has no ctypes signature"
    assert set(nat.SIGNATURES) <= set(names)
    assert lib.octl_abi_version() == 1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_product_path_fails_loudly_without_gpu():
    from octreelib_amd import _native as nat
    from octreelib_amd.grid import Grid, GridConfig
    from octreelib_amd.ransac import CudaRansac

    with pytest.raises(nat.NativeLibraryError):
        Grid(GridConfig(voxel_edge_length=1))
    with pytest.raises(nat.NativeLibraryError):
        CudaRansac().evaluate(np.zeros((8, 3)), np.array([8], dtype=np.int32))


def test_grid_config_type_errors_verbatim():
    # test/grid/test_grid.py:148-182 of the reference
    from octreelib_amd.grid import GridConfig
    from octreelib_amd.octree import OctreeConfig
    from octreelib_amd.octree_manager import OctreeManager

    with pytest.raises(TypeError) as e:
        GridConfig(octree_manager_type=type(None), octree_config=OctreeConfig(), voxel_edge_length=5)
    assert str(e.value) == (
        "Cannot use the provided octree manager type NoneType. "
        "It has to be a subclass of octree_manager.OctreeManager."
    )
    with pytest.raises(TypeError) as e:
        GridConfig(octree_manager_type=OctreeManager, octree_type=type(None), octree_config=OctreeConfig(),
                   voxel_edge_length=5)
    assert str(e.value) == (
        "Cannot use the provided octree type NoneType. "
        "It has to be a subclass of octree.OctreeBase."
    )


def test_count_criterion_recognition():
    from octreelib_amd.criteria import MaxPoints, UnsupportedCriterion, count_threshold

    k = 12
    assert count_threshold([lambda points: len(points) > 2]) == 2
    assert count_threshold([lambda pts: len(pts) > k]) == 12
    assert count_threshold([lambda p: len(p) >= 5]) == 4
    assert count_threshold([lambda p: 7 < len(p)]) == 7
    assert count_threshold([MaxPoints(9), lambda p: len(p) > 30]) == 9
    assert count_threshold([]) == -1

    def named(points):
        return len(points) > 3

    assert count_threshold([named]) == 3
    with pytest.raises(UnsupportedCriterion):
        count_threshold([lambda p: p[:, 2].std() > 0.1])
    with pytest.raises(UnsupportedCriterion):
        count_threshold([lambda p: len(p) > 2 and True])
    with pytest.raises(RecursionError):
        count_threshold([lambda p: len(p) > -1])


def test_filter_interval_recognition_handles_non_finite_constants_and_probes_both_ends():
    from octreelib_amd.criteria import try_count_interval

    assert try_count_interval([lambda p: len(p) < 100]) == (0, 99)
    assert try_count_interval([lambda p: len(p) >= 2, lambda p: len(p) <= 50.5]) == (2, 50)
    assert try_count_interval([lambda p: 5 <= len(p)])[0] == 5
    assert try_count_interval([lambda p: len(p) == 3]) == (3, 3)
    # non-finite constants: no crash, the host path takes them
    assert try_count_interval([lambda p: len(p) < float("inf")]) is None
    assert try_count_interval([lambda p: len(p) > np.nan]) is None
    assert try_count_interval([lambda p: len(p) > -np.inf]) is None

    # a function whose bytecode looks like `len(p) < c` but whose upper end behaves differently is
    # caught by the probe at hi (c changes between recognition and the probe calls)
    class Shifty:
        def __init__(self):
            self.calls = 0

        def __float__(self):
            return 100.0

        def __gt__(self, n):  # len(p) < self  ->  self > len(p)
            return n < 50

        __rlt__ = __gt__

    shifty = Shifty()
    assert try_count_interval([lambda p: len(p) < shifty]) is None

    # the same mismatch far above any cloud one would allocate: the probes are zero-stride views, so both ends of
    # an interval of ANY size are checked behaviourally (a bound above 4 M used to be accepted on the pattern alone)
    class ShiftyBig(Shifty):
        def __float__(self):
            return 1e10

        def __gt__(self, n):
            return n < 5_000_000_000

        __rlt__ = __gt__

    big = ShiftyBig()
    assert try_count_interval([lambda p: len(p) < big]) is None
    assert try_count_interval([lambda p: len(p) < 10_000_000_000]) == (0, 9_999_999_999)
    from octreelib_amd.criteria import count_threshold

    assert count_threshold([lambda p: len(p) > 8_000_000_000]) == 8_000_000_000


def test_voxel_value_type():
    # reference: internal/voxel.py - equal voxels share an id, hash/eq on (corner, edge)
    from octreelib_amd.internal import Voxel, VoxelBase

    a = Voxel(np.array([0, 0, 2.5]), 2.5, np.zeros((2, 3)))
    b = VoxelBase(np.array([0.0, 0.0, 2.5]), 2.5)
    c = Voxel(np.array([0, 0, 2.5]), 1.25)
    assert a == b and hash(a) == hash(b) and a.id == b.id
    assert a != c and a.id != c.id
    assert np.array_equal(a.corner_max, np.array([2.5, 2.5, 5.0]))
    assert len(a.all_corners) == 8
    a.insert_points(np.ones((3, 3)))
    assert a.get_points().shape == (5, 3)


def test_voxel_owner_host_mirror_matches_library():
    from octreelib_amd import _native as nat
    from octreelib_amd.distributed import voxel_owner_np

    lib = nat.load()
    rng = np.random.default_rng(0)
    q = rng.integers(-1000, 1000, (2000, 3))
    for n in (1, 2, 3, 8):
        want = np.array([lib.octl_voxel_owner(int(a), int(b), int(c), n) for a, b, c in q])
        assert np.array_equal(voxel_owner_np(q, n), want)
    # reasonably balanced over 8 ranks
    counts = np.bincount(voxel_owner_np(np.argwhere(np.ones((32, 32, 32))), 8), minlength=8)
    assert counts.min() > 0.9 * counts.mean()


def test_bench_gpus_n_starts_its_own_ranks_or_says_why_not():
    """`python bench.py --gpus N` without a launcher must start N ranks itself - before the parent touches
    a GPU - and, on a machine with fewer GPUs, exit 2 with a clear message instead of running one rank."""
    import subprocess
    import sys

    import bench

    have = bench.visible_gpus()
    if have >= 2:
        pytest.skip("this machine could really run two ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert out.returncode == 2
    assert "needs 2 GPUs" in out.stderr and out.stdout.strip() == ""


def test_bench_plan_touches_no_gpu_and_fits_the_drivers_budget():
    """`bench.py --gpus 8 --plan`: what the 8-rank run of BASELINE config 5 needs, printed as one JSON object without
    starting a rank or touching a GPU (it must run in the build container)."""
    import json
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plan"], capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["ranks"] == 8 and d["points_per_rank"] == 125_000_000 and d["total_points"] == 10 ** 9
    assert d["device"]["hbm_GB_per_rank_estimate"] < d["device"]["hbm_GB_available"]
    assert d["exchange"]["bytes_per_point"] == 32 and d["exchange"]["MB_per_peer_message"] == 500.0
    assert d["seconds"]["wall_estimate"] < d["seconds"]["driver_budget"] == 600 and d["fits_driver_budget"] is True
    # the strong-scaling series divides the N = 1 cloud
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plan", "--scaling", "strong"],
                         capture_output=True, text=True, timeout=120)
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["ranks"] == 4 and d["points_per_rank"] == 2_500_000 and d["resident_clouds_per_rank"] == 3


def test_key_geometry_pads_the_box_and_divides_exactly():
    """The geometry of a single-pass bucket build (csrc/bucket_build.hip: geom_from_box, the same code on the host and
    on the device) through its host-side test hook: the keys' box is the true box padded by the margin, buckets are
    runs of `width` keys with no more buckets than the tables hold, and the kernels' division-free bucket index is
    key / width for EVERY key of the box (grid.py:72-90 rebuckets from scratch per call; here a geometry must only
    depend on the true box)."""
    import ctypes as C

    from octreelib_amd import _native as nat

    lib = nat.load()
    rng = np.random.default_rng(0)
    cases = [((0, 0, 0, 31, 31, 31), 4096, 10_000_000, 1), ((0, 0, 0, 6, 6, 6), 64, 100_000, 1),
             ((-3, 5, 100, 4, 9, 131), 256, 600_000, 2), ((0, 0, 0, 0, 0, 0), 1, 50, 1), ((7, 7, 7, 7, 7, 7), 8, 20_000, 0),
             ((0, 0, 0, 255, 255, 31), 4096, 10_000_000, 1), ((0, 0, 0, 31, 31, 31), 4096, 10_000_000, 0)]
    for _ in range(40):
        lo = rng.integers(-50, 50, 3)
        ext = rng.integers(1, 40, 3)
        n = int(rng.integers(1_000, 20_000_000))
        want = 1
        while want < 4096 and want * 2560 < n:
            want *= 2
        cases.append((tuple(int(v) for v in lo) + tuple(int(v) for v in lo + ext - 1), want, n, int(rng.integers(0, 4))))
    seen_valid = 0
    for tb, want, n, margin in cases:
        tb_a = (C.c_int32 * 6)(*tb)
        bb = (C.c_int32 * 6)()
        width, nb, valid, bad = C.c_uint32(0), C.c_uint32(0), C.c_int32(0), C.c_int64(-1)
        rc = lib.octl_debug_key_geometry(C.cast(tb_a, C.c_void_p), want, n, 2560, margin, C.cast(bb, C.c_void_p),
                                         C.byref(width), C.byref(nb), C.byref(valid), C.byref(bad))
        assert rc == 0
        if valid.value != 1:
            assert valid.value == -3       # GEOM_RETRY: not a single-pass case (too many keys for the tables)
            continue
        seen_valid += 1
        assert list(bb) == [tb[0] - margin, tb[1] - margin, tb[2] - margin, tb[3] + margin, tb[4] + margin, tb[5] + margin]
        cap = min(4096, max(64, 2 * want))
        assert 1 <= width.value <= 4096 and 1 <= nb.value <= cap, (tb, want, width.value, nb.value)
        assert bad.value == 0, (tb, width.value, bad.value)
        # about `target` points per bucket of a full box, unless the tables force wider buckets
        rt = (tb[3] - tb[0] + 1) * (tb[4] - tb[1] + 1) * (tb[5] - tb[2] + 1)
        rp = (bb[3] - bb[0] + 1) * (bb[4] - bb[1] + 1) * (bb[5] - bb[2] + 1)
        assert width.value == max(1, (2560 * rt + n // 2) // n, -(-rp // cap))
    assert seen_valid >= 30
    # the headline scene: 32^3 voxels, one voxel of slack -> 34^3 keys in buckets of 10
    tb_a = (C.c_int32 * 6)(0, 0, 0, 31, 31, 31)
    lib.octl_debug_key_geometry(C.cast(tb_a, C.c_void_p), 4096, 10_000_000, 2560, 1, C.cast(bb, C.c_void_p),
                                C.byref(width), C.byref(nb), C.byref(valid), C.byref(bad))
    assert (width.value, nb.value, valid.value, bad.value) == (10, 3931, 1, 0)
