"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (octreelib_amd/).

NumPy restatement of the reference's per-leaf RANSAC kernel, vectorised over the
hypotheses of one block (reference: one CUDA block per leaf, one thread per hypothesis).
Restates ransac/cuda_ransac.py:85-155 and ransac/util.py:12-84 of /root/reference with
the arithmetic of the CI path (numba CUDA *simulator*: IEEE f64, no FMA contraction),
evaluated in the reference's operation order.

Deterministic choices where the reference is not:
  * winner among hypotheses tied at the block maximum: the reference lets any of them win
    (CAS race, cuda_ransac.py:140-145); here the LOWEST hypothesis index wins.  The set of
    tied planes is returned too so a reference run can be checked against it.
  * ``norm`` is ``sqrt(a*a+b*b+c*c)``; the reference writes ``(a**2+b**2+c**2) ** 0.5``
    (util.py:76), i.e. libm ``pow`` under the simulator — equal after the f32 store except
    for astronomically rare double-rounding cases; pinned by tests/golden/ransac_*.npz
    which come from the reference kernel source itself.
  * a sampled index that lands past the end of the whole cloud (possible only through
    f64 rounding of ``R*n + s`` in the last block) is clamped to the last point; the
    reference would read out of bounds.
"""

import numpy as np


def plane_from_points(pts: np.ndarray):
    """util.py:27-84.  pts: (H, k, 3) f64 sampled points -> (H, 4) f64 plane."""
    H, k, _ = pts.shape
    cx = np.zeros(H)
    cy = np.zeros(H)
    cz = np.zeros(H)
    for i in range(k):  # util.py:37-40 sequential sums
        cx = cx + pts[:, i, 0]
        cy = cy + pts[:, i, 1]
        cz = cz + pts[:, i, 2]
    cx = cx / k
    cy = cy / k
    cz = cz / k
    xx = np.zeros(H)
    xy = np.zeros(H)
    xz = np.zeros(H)
    yy = np.zeros(H)
    yz = np.zeros(H)
    zz = np.zeros(H)
    for i in range(k):  # util.py:48-57
        rx = pts[:, i, 0] - cx
        ry = pts[:, i, 1] - cy
        rz = pts[:, i, 2] - cz
        xx = xx + rx * rx
        xy = xy + rx * ry
        xz = xz + rx * rz
        yy = yy + ry * ry
        yz = yz + ry * rz
        zz = zz + rz * rz
    det_x = yy * zz - yz * yz
    det_y = xx * zz - xz * xz
    det_z = xx * yy - xy * xy
    bx = (det_x > det_y) & (det_x > det_z)  # util.py:63
    by = ~bx & (det_y > det_z)  # util.py:67
    bz = ~bx & ~by
    ax = np.where(bx, det_x, np.where(by, xz * yz - xy * zz, xy * yz - xz * yy))
    ay = np.where(bx, xz * yz - xy * zz, np.where(by, det_y, xy * xz - yz * xx))
    az = np.where(bx, xy * yz - xz * yy, np.where(by, xy * xz - yz * xx, det_z))
    del bz
    norm = np.sqrt(ax * ax + ay * ay + az * az)
    zero = norm == 0
    safe = np.where(zero, 1.0, norm)
    ax = ax / safe
    ay = ay / safe
    az = az / safe
    d = -(ax * cx + ay * cy + az * cz)
    plane = np.stack([ax, ay, az, d], axis=1)
    plane[zero] = 0.0  # util.py:77-78
    return plane


def evaluate(point_cloud, block_sizes, hypotheses, threshold, details=False):
    """
    cuda_ransac.py:43-155.
    point_cloud (M,3) f64, block_sizes (B,) int32, hypotheses (H,k) f64 in [0,1).
    Returns mask (M,) bool; with details=True also best_count (B,) i32,
    best_plane (B,4) f32 (lowest-index maximiser), best_index (B,) i32 and a list of the
    unique tied f32 planes per block.
    """
    cloud = np.ascontiguousarray(point_cloud, dtype=np.float64).reshape(-1, 3)
    sizes = np.asarray(block_sizes, dtype=np.int32)
    R = np.asarray(hypotheses, dtype=np.float64)
    H, k = R.shape
    M = len(cloud)
    B = len(sizes)
    starts = np.cumsum(np.concatenate(([0], sizes[:-1]))).astype(np.int64)  # :64-66
    mask = np.zeros(M, dtype=np.bool_)
    best_count = np.zeros(B, dtype=np.int32)
    best_plane = np.zeros((B, 4), dtype=np.float32)
    best_index = np.full(B, -1, dtype=np.int32)
    tied = [None] * B
    thr = np.float64(threshold)
    for b in range(B):
        n = sizes[b]
        s = starts[b]
        if n < k:  # :96-97
            continue
        idx = (R * n + s).astype(np.int32)  # :103-107 (f64 multiply, f64 add, truncation)
        idx = np.minimum(idx, M - 1)
        plane32 = plane_from_points(cloud[idx]).astype(np.float32)  # :110-113
        p = plane32.astype(np.float64)
        blk = cloud[s : s + n]
        # util.py:22-24 — ((a*x + b*y) + c*z) + d, f32 plane promoted to f64
        dist = np.abs(
            (
                (p[:, 0:1] * blk[None, :, 0] + p[:, 1:2] * blk[None, :, 1])
                + p[:, 2:3] * blk[None, :, 2]
            )
            + p[:, 3:4]
        )
        counts = (dist < thr).sum(axis=1).astype(np.int32)  # :116-121
        mx = counts.max()
        win = int(np.argmax(counts == mx))  # lowest index among the tied
        best_count[b] = mx
        best_plane[b] = plane32[win]
        best_index[b] = win
        if details:
            tied[b] = np.unique(plane32[counts == mx], axis=0)
        mask[s : s + n] = dist[win] < thr  # :149-155
    if details:
        return mask, best_count, best_plane, best_index, tied
    return mask
