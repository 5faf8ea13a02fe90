#!/usr/bin/env python3
"""Phase clocks of k_bucket_build (a library built with -DBB_STAMPS, see csrc/bucket_build.hip):
   tools/build_variant.sh stamps "-DBB_STAMPS" && OCTREELIB_AMD_LIB=build/variants/stamps.so python tools/bb_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat

ctx = nat.Context(0)
lib = ctx.lib
lib.octl_debug_bb_stamps.restype = C.c_int
lib.octl_debug_bb_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
# (points, voxels per side, buckets of the build: the headline by default; `100000 7 64` = the dense small scan)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
SIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 32
NBK = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
wl = bench.Workload(ctx, ctx, 0, 1, N, (SIDE, SIDE, SIDE), "planar", 64, False, False, n_clouds=1)
for _ in range(3):
    wl.step_build_only()
out = (C.c_ulonglong * 16)()
ctx.check(lib.octl_debug_bb_stamps(ctx.handle, out, 1))
reps = 5
for _ in range(reps):
    wl.step_build_only()
ctx.check(lib.octl_debug_bb_stamps(ctx.handle, out, 0))
names = ["load record tails", "level pyramid", "keys + sort", "outputs"]
tot = sum(out[i] for i in range(4))
print("k_bucket_build")
for i, nm in enumerate(names):
    print("  %-28s %6.1f %%   %8.0f cycles per bucket" % (nm, 100.0 * out[i] / tot, out[i] / reps / NBK))
print("  total cycles per bucket: %.0f" % (tot / reps / NBK))
names = ["bases + roots", "internal nodes + children", "preorder ranks", "blocks", "block order"]
tot = sum(out[4 + i] for i in range(5))
print("k_bucket_finish")
for i, nm in enumerate(names):
    print("  %-28s %6.1f %%   %8.0f cycles per bucket" % (nm, 100.0 * out[4 + i] / tot, out[4 + i] / reps / NBK))
print("  total cycles per bucket: %.0f" % (tot / reps / NBK))
names = ["counters reset", "loads, keys, ranks", "wait for the other waves", "wave offsets", "stores issued", "barrier behind the stores"]
tot = sum(out[9 + i] for i in range(6))
TILES = -(-N // (256 * 8))   # PT_THREADS x OCTL_PT_IPT records per tile
print("k_part_scatter (cycles per %d-record tile)" % (256 * 8))
for i, nm in enumerate(names):
    print("  %-28s %6.1f %%   %8.0f" % (nm, 100.0 * out[9 + i] / max(tot, 1), out[9 + i] / reps / TILES))
print("  total: %.0f" % (tot / reps / TILES))
print("  (of 'loads, keys, ranks': waiting for the tile's loads %.0f)" % (out[15] / reps / TILES))
wl.close()
