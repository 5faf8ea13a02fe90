import sys, numpy as np
sys.path.insert(0, '.')
from octreelib_amd import synthetic
from octreelib_amd._engine import Forest
from octreelib_amd.ransac import CudaRansac
pts = synthetic.planar_cloud(2_000_000, (19, 19, 19), seed=1)
f = Forest(0, np.zeros(3), 1.0); f.add_pose(pts); f.subdivide(64)
blk = f.blocks; xyz = f.xyz
sizes = blk["size"].astype(np.int32)
np.random.seed(0)
op = CudaRansac(0.01, 1024, 6)
mask, planes, counts, index = op.evaluate(xyz, sizes, details=True)
ev = sizes >= 6
perfect = ev & (counts == sizes)
print("evaluated leaves", ev.sum(), "perfect", perfect.sum(), "fraction %.3f" % (perfect.sum() / ev.sum()))
print("winner index among perfect: median", np.median(index[perfect]), "frac < 64:", (index[perfect] < 64).mean(), "< 256:", (index[perfect] < 256).mean())
print("fits saved if groups after the first perfect one are skipped (per-wave view ignored): %.3f" % (((1024 - 256 * (index[perfect] // 256 + 1)) / 1024).sum() / ev.sum()))
w = sizes[ev].astype(float)
print("size-weighted perfect fraction %.3f" % (sizes[perfect].sum() / sizes[ev].sum()))
# estimated kernel cost under a size-gated two-pass policy (VALU instructions per wavefront and leaf:
# 4 fits x 267, scoring 22 n, 150 other; two-pass structure +6 %; exit saves 3 fits + 3/4 of the scoring)
n_ev = sizes[ev].astype(float); perf = perfect[ev] & (index[ev] < 256)
base = (1068 + 22 * n_ev + 150)
print("baseline total", base.sum())
for T in (8, 12, 16, 20, 24, 32, 48, 64, 255):
    two = n_ev <= T
    cost = np.where(two, base * 1.06 - np.where(perf, 801 + 16.5 * n_ev, 0.0), base)
    print("two-pass for n <= %3d: leaves %.2f, perfect among them %.2f, cost ratio %.4f" % (T, two.mean(), perf[two].mean() if two.any() else 0, cost.sum() / base.sum()))
