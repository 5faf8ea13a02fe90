#!/usr/bin/env python3
"""What k_ransac EXECUTES of what the algorithm asks for (a library built with -DRS_COUNTS, see csrc/ransac.hip):
   tools/build_variant.sh rs_counts "-DRS_COUNTS" && OCTREELIB_AMD_LIB=build/variants/rs_counts.so python tools/rs_counts.py
Counts per launch of the benchmarked instance (one wave per block): blocks, blocks that leave after pass 1 (256
hypotheses) or after a later batch, plane fits executed of the 1024 per block the reference runs, (point, hypothesis)
pairs scored by the f32 screen, hypotheses recounted in f64.  -> gpurun_out/rs_counts.json (kept as
profiles/rNN_ransac_counts.json; bench.py's roofline_valu.executed reads it)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from octreelib_amd import _native as nat

ctx = nat.Context(0)
lib = ctx.lib
lib.octl_debug_rs_stamps.restype = C.c_int
lib.octl_debug_rs_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
wl = bench.Workload(ctx, ctx, 0, 1, 10_000_000, (32, 32, 32), "planar", 64, False, False, n_clouds=3)
for _ in range(3):
    wl.step()
out = (C.c_ulonglong * 16)()
ctx.check(lib.octl_debug_rs_stamps(ctx.handle, out, 1))
reps = 6
for _ in range(reps):
    wl.step()
ctx.sync()
ctx.check(lib.octl_debug_rs_stamps(ctx.handle, out, 0))
v = [out[i] / reps for i in range(16)]
blocks, exit0, pts = v[8], v[9], v[11]
H, LANES, HPL = 1024, 64, 16
eligible, a_groups, survivors, b_batches, overflow = v[4], v[12], v[2], v[3], v[5]
exact_groups = v[14]           # hypothesis groups of 64 fitted and scored exactly (group 0 + the batches of survivors)
res = {
    # the kernel's own clock: shader-clock ticks over 100 MHz ticks, summed over the workgroups' lives
    "in_kernel_clock_GHz": (v[0] / v[1]) * 0.1 if v[1] else None,
    "instance": "k_ransac<64,16,6,0,true,true> (blocks of 6..63 points; the 64-point leaves run in k_ransac<128,8,...> "
                "and are not counted here)",
    "blocks_per_launch": blocks, "mean_block_size": pts / blocks,
    "blocks_leaving_after_group_0": exit0, "fraction_blocks_leaving_after_group_0": exit0 / blocks,
    "blocks_with_prescreen": eligible, "fraction_blocks_with_prescreen": eligible / blocks,
    "blocks_whose_survivors_overflowed_the_queue": overflow,
    "hypotheses_prescreened": a_groups * LANES, "hypotheses_prescreened_per_block": a_groups * LANES / max(eligible, 1),
    "survivors": survivors, "survivors_per_prescreened_block": survivors / max(eligible, 1),
    "fraction_prescreened_that_survive": survivors / max(a_groups * LANES, 1),
    "survivor_batches": b_batches, "survivor_batches_per_block": b_batches / blocks,
    "plane_fits_asked": blocks * H, "plane_fits_executed_exactly": exact_groups * LANES,
    "fraction_plane_fits_executed_exactly": exact_groups / (blocks * HPL),
    "fraction_plane_fits_executed": exact_groups / (blocks * HPL),   # (the key bench.py reads)
    "pairs_asked": pts * H, "pairs_screened_f32_exact_path": v[15] * LANES,
    "fraction_pairs_scored": v[15] * LANES / (pts * H),
    "hypotheses_recounted_f64": v[13], "fraction_hypotheses_recounted": v[13] / max(exact_groups * LANES, 1),
    "fraction_blocks_leaving_after_pass_1": exit0 / blocks,          # (the key bench.py reads)
    "note": "average per launch over %d steps of the rotating headline workload; the instance runs one wave per block" % reps,
}
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/rs_counts.json", "w"), indent=1)
wl.close()
