#!/usr/bin/env python3
"""Soak of the speculative k_bucket_finish (run on the GPU box): a random sequence of scans - sizes from 2 k to 1.5 M
points, dense, thin and skewed scenes, K from 4 to 200, fresh forests on one context - each built with the speculative
launch allowed and again with NO_SPEC_FINISH, without the geometry hint, and under another margin of the key geometry;
scheme, blocks, order, permutation, RANSAC mask and compaction must be identical.  usage: tools/soak_spec.py [iterations] [seed]"""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np
import tests.test_gpu_parity as T
from octreelib_amd import _native as nat
from octreelib_amd import synthetic

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
ctx = nat.get_context()


def counters():
    h, m = C.c_uint64(0), C.c_uint64(0)
    ctx.check(ctx.lib.octl_debug_spec_finish(C.byref(h), C.byref(m)))
    return h.value, m.value


bad = 0
numbering_only = 0
h0, m0 = counters()
for it in range(iters):
    n = int(10 ** rng.uniform(3.3, 6.2))
    kind = rng.integers(0, 3)
    side = int(rng.integers(2, 20))
    K = int(rng.choice([4, 16, 32, 64, 200]))
    if kind == 2:
        pts = synthetic.sparse_scene(n, (side + 4, side + 4, max(2, side // 2)), seed=int(rng.integers(1, 1000)),
                                     cluster_fraction=0.3, cluster_density=float(rng.choice([20.0, 60.0])))
    else:
        pts = synthetic.planar_cloud(n, (side, side, side), seed=int(rng.integers(1, 1000)), stream=int(rng.integers(1, 50)))
    if rng.integers(0, 4) == 0:
        pts = pts + np.array([float(rng.integers(0, 5)), 0.0, float(rng.integers(0, 3))])   # (another voxel box)
    adopt = ctx if rng.integers(0, 2) else None
    def leg(name, want, got, canonical):
        """0 = equal; a difference that is only the numbering of the nodes (benign, see below) is reported apart"""
        global numbering_only
        try:
            T._assert_same_step(want, got, canonical=canonical)
        except AssertionError as e:
            if not canonical:
                try:
                    T._assert_same_step(want, got, canonical=True)
                    numbering_only += 1
                    print("numbering only at", it, name, n, kind, side, K, repr(e)[:80], flush=True)
                    return
                except AssertionError:
                    pass
            raise AssertionError(name + ": " + repr(e)[:160])

    try:
        got = T._step_tables([pts], K, adopt)
        ctx.set_option("NO_SPEC_FINISH", 1)
        want = T._step_tables([pts], K, adopt)
        ctx.set_option("NO_SPEC_FINISH", 0)
        # (the two builds may run under different geometry hints - the first under the previous scan's: where the
        #  buckets are cut elsewhere on a skewed scene, other voxels are left to the level loop, which numbers their
        #  nodes behind the others - same trees, same leaves, same order, other ids)
        leg("spec", want, got, False)
        # round 6: the same scan without the geometry hint of the context's previous build (its own box pass), and
        # under another margin of the key geometry - the tables never depend on either
        ctx.set_option("NO_GEOM_HINT", 1)
        leg("no hint", want, T._step_tables([pts], K, adopt), True)
        ctx.set_option("NO_GEOM_HINT", 0)
        ctx.set_option("GEOM_MARGIN", int(rng.choice([-1, 2, 3])))
        leg("margin", want, T._step_tables([pts], K, adopt), True)
        ctx.set_option("GEOM_MARGIN", 0)
        # the small-launch forms (k_block_prepare_small, k_mask_scan's own compaction, the scans inside
        # k_bucket_finish<true> and k_part_scatter<..., true>) against the separate kernels, and another bucket size
        ctx.set_option("NO_FUSED_TABLES", 1)
        leg("unfused", want, T._step_tables([pts], K, adopt), False)
        ctx.set_option("NO_FUSED_TABLES", 0)
        ctx.set_option("BUCKET_POINTS", int(rng.choice([640, 2560, 5000])))
        leg("bucket size", want, T._step_tables([pts], K, adopt), True)
        ctx.set_option("BUCKET_POINTS", 0)
    except Exception as e:  # noqa: BLE001
        bad += 1
        ctx.set_option("NO_SPEC_FINISH", 0)
        ctx.set_option("NO_GEOM_HINT", 0)
        ctx.set_option("GEOM_MARGIN", 0)
        ctx.set_option("NO_FUSED_TABLES", 0)
        ctx.set_option("BUCKET_POINTS", 0)
        print("FAILED at", it, n, kind, side, K, repr(e)[:200], flush=True)
    if it % 20 == 19:
        print("iteration", it + 1, "speculative launches held / missed so far:", tuple(a - b for a, b in zip(counters(), (h0, m0))),
              flush=True)
print("failures:", bad, "| differences in the numbering of the nodes only:", numbering_only)
sys.exit(1 if bad else 0)
