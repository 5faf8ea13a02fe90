#!/usr/bin/env python3
"""Cycle ledger of k_ransac (a library built with -DRS_STAMPS, see csrc/ransac.hip):
   tools/build_variant.sh rs_stamps "-DRS_STAMPS" && OCTREELIB_AMD_LIB=build/variants/rs_stamps.so python tools/rs_stamps.py
Wave 0 of every workgroup clocks the phases of every block; the shares are applied to the kernel's measured time and
set beside the static instruction counts of the same phases (tools/ransac_isa.py)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from octreelib_amd import _native as nat

WAVES = 1   # waves per block of the benchmarked instance, k_ransac<64,16,6>
ctx = nat.Context(0)
lib = ctx.lib
lib.octl_debug_rs_stamps.restype = C.c_int
lib.octl_debug_rs_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
wl = bench.Workload(ctx, ctx, 0, 1, 10_000_000, (32, 32, 32), "planar", 64, False, False)
for _ in range(3):
    wl.step()
out = (C.c_ulonglong * 16)()
ctx.check(lib.octl_debug_rs_stamps(ctx.handle, out, 1))
reps = 5
ctx.set_profiling(2)
for _ in range(reps):
    wl.step()
ctx.sync()
tm = ctx.timings()
ctx.set_profiling(False)
ctx.check(lib.octl_debug_rs_stamps(ctx.handle, out, 0))
ms = tm["ransac"][0] / tm["ransac"][1]
names = ["iteration head (prefetch, sample positions)", "plane fits, pass 1 (256 hypotheses)", "scoring, pass 1",
         "plane fits, pass 2 (768 hypotheses)", "scoring, pass 2", "reduction + staging of the next block",
         "barrier", "winner, outputs, final mask"]
tot = sum(out[i] for i in range(8))
blocks, skipped, wgs, pts = out[8] / reps, out[9] / reps, out[10] / reps, out[11] / reps
res = {"kernel_ms": ms, "blocks": blocks, "blocks_without_pass_2": skipped, "workgroups": wgs,
       "mean_block_size": pts / blocks, "phases": {}}
print("k_ransac %.3f ms (this build, stamps included); %d blocks (%.1f %% exit after pass 1), mean size %.1f" % (
    ms, blocks, 100.0 * skipped / blocks, pts / blocks))
for i, nm in enumerate(names):
    share = out[i] / tot
    res["phases"][nm] = {"share": share, "ms": share * ms, "clocks_per_block": out[i] / reps / blocks}
    print("  %-46s %5.1f %%  %6.3f ms  %9.0f clocks per block" % (nm, 100 * share, share * ms, out[i] / reps / blocks))
fits = blocks * 256 + (blocks - skipped) * 768
print("  plane fits per launch: %.3e (%.0f wave-fits); clocks per wave-fit: %.1f" % (
    fits, fits / 64, (out[1] + out[3]) / reps / (fits / 64 / WAVES)))   # wave 0 executes 1 / WAVES of a block's wave-fits
res["fits_per_launch"] = fits
json.dump(res, open("gpurun_out/rs_stamps.json", "w"), indent=1)
wl.close()
