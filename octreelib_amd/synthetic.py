"""
Synthetic clouds of the benchmark configurations: uniform and
"planar" (per top-level voxel 80 % of the points on a random plane + noise, 20 % uniform
outliers; shape follows the reference's own test generator, test/grid/test_cuda_ransac.py:9-24).
"""

import numpy as np

__all__ = ["uniform_cloud", "planar_cloud", "sweep_order"]


def uniform_cloud(n: int, dims=(32, 32, 32), seed: int = 0, voxels=None) -> np.ndarray:
    """n uniform points over the scene; `voxels` (linear ids, x slowest) restricts them to those voxels."""
    rng = np.random.default_rng(seed)
    if voxels is None:
        return rng.random((n, 3)) * np.asarray(dims, dtype=np.float64)
    dims = np.asarray(dims, dtype=np.int64)
    lin = np.asarray(voxels, dtype=np.int64)[rng.integers(0, len(voxels), n)]
    q = np.stack([lin // (dims[1] * dims[2]), (lin // dims[2]) % dims[1], lin % dims[2]], axis=1)
    return rng.random((n, 3)) + q.astype(np.float64)


def planar_cloud(n: int, dims=(32, 32, 32), seed: int = 1, stream: int = 0,
                 inlier_fraction: float = 0.8, sigma: float = 0.005, box=None, voxels=None) -> np.ndarray:
    """n points over a grid of dims[0] x dims[1] x dims[2] voxels of 1 m.  The plane of every
    voxel depends only on `seed`, so different `stream`s (ranks) draw different points of the
    SAME scene.  `box` = (lo, hi) integer voxel bounds restricts the points to a sub-box;
    `voxels` (linear ids, x slowest) to a set of voxels (e.g. the voxels one rank owns)."""
    dims = np.asarray(dims, dtype=np.int64)
    V = int(dims.prod())
    ab = np.random.default_rng(seed).uniform(-0.4, 0.4, (V, 2))
    rng = np.random.default_rng([seed, stream, 0x5EED])
    if box is None:
        lo, hi = np.zeros(3, dtype=np.int64), dims
    else:
        lo, hi = np.asarray(box[0], dtype=np.int64), np.asarray(box[1], dtype=np.int64)
    if voxels is not None:
        lin = np.asarray(voxels, dtype=np.int64)[rng.integers(0, len(voxels), n)]
        q = np.stack([lin // (dims[1] * dims[2]), (lin // dims[2]) % dims[1], lin % dims[2]], axis=1)
    else:
        q = np.stack([rng.integers(lo[a], hi[a], n) for a in range(3)], axis=1)
        lin = (q[:, 0] * dims[1] + q[:, 1]) * dims[2] + q[:, 2]
    local = rng.random((n, 3))
    inl = rng.random(n) < inlier_fraction
    a, b = ab[lin, 0], ab[lin, 1]
    z = 0.5 + a * (local[:, 0] - 0.5) + b * (local[:, 1] - 0.5) + rng.normal(0.0, sigma, n)
    local[inl, 2] = np.clip(z[inl], 1e-9, 1.0 - 1e-9)
    return local + q.astype(np.float64)


def sparse_scene(n: int, dims=(256, 256, 32), seed: int = 7, cluster_fraction: float = 0.03,
                 cluster_density: float = 20.0) -> np.ndarray:
    """A scene that is NOT dense in its bounding box - what a real scan looks like: a terrain sheet about 1.6
    voxels thick through a dims[0] x dims[1] x dims[2] box of 1 m voxels (about 5 % of the voxels occupied,
    ~95 points per occupied voxel at n = 10 M), plus one blob that holds `cluster_fraction` of the points at
    `cluster_density` times the sheet's density."""
    rng = np.random.default_rng(seed)
    dims = np.asarray(dims, dtype=np.float64)
    n_cl = int(n * cluster_fraction)
    m = n - n_cl
    x = rng.random(m) * dims[0]
    y = rng.random(m) * dims[1]
    zc = 0.5 * dims[2] + 0.45 * dims[2] * np.sin(x / 40.0) * np.cos(y / 55.0)
    z = zc + (rng.random(m) - 0.5) * 1.6
    sheet = np.stack([x, y, np.clip(z, 0.0, dims[2] - 1e-9)], axis=1)
    # the blob: density = cluster_density x (points per unit volume of the sheet)
    sheet_density = m / (dims[0] * dims[1] * 1.6)
    vol = n_cl / (cluster_density * sheet_density)
    r = vol ** (1.0 / 3.0)
    c = np.array([0.37 * dims[0], 0.61 * dims[1], 0.5 * dims[2]])
    blob = c + (rng.random((n_cl, 3)) - 0.5) * r
    pts = np.vstack([sheet, blob])
    rng.shuffle(pts)
    return np.ascontiguousarray(pts)


def sweep_order(points: np.ndarray, seed: int = 0, run_min: int = 8, run_max: int = 64, edge: float = 1.0) -> np.ndarray:
    """The same points in the order a rotating LiDAR delivers them: RUNS of run_min .. run_max consecutive points that
    fall into one top-level voxel (a beam sweeping over a surface), the runs themselves in random order.  (The
    benchmark's clouds are shuffled point by point - the partition's worst case; the reference's own generator emits
    voxel after voxel, test/grid/test_cuda_ransac.py:9-24.)"""
    pts = np.asarray(points, dtype=np.float64)
    n = len(pts)
    rng = np.random.default_rng(seed)
    q = np.floor(pts / edge).astype(np.int64)
    q -= q.min(axis=0)
    dims = q.max(axis=0) + 1
    lin = (q[:, 0] * dims[1] + q[:, 1]) * dims[2] + q[:, 2]
    by_voxel = np.argsort(lin, kind="stable")
    lin_sorted = lin[by_voxel]
    # cut the voxel-sorted sequence into runs: at every voxel boundary and after a random length inside a voxel
    cut = np.zeros(n + 1, dtype=bool)
    cut[0] = cut[n] = True
    cut[1:n] = lin_sorted[1:] != lin_sorted[:-1]
    pos = 0
    lengths = rng.integers(run_min, run_max + 1, size=n // run_min + 2)
    marks = np.cumsum(lengths)
    marks = marks[marks < n]
    cut[marks] = True        # (a run may end early at a voxel boundary: still between 1 and run_max points of one voxel)
    starts = np.flatnonzero(cut[:-1])
    ends = np.flatnonzero(cut[1:]) + 1
    order = rng.permutation(len(starts))
    idx = np.concatenate([by_voxel[starts[r]:ends[r]] for r in order]) if n else by_voxel
    return np.ascontiguousarray(pts[idx])
