#!/usr/bin/env python3
"""Timeline of one step of the headline out of a rocprofv3 --kernel-trace CSV: start, duration, gap to the end of
everything before it, stream.   usage: tools/probes/step_timeline.py <dir with *_kernel_trace.csv> [planar|sparse]"""
import csv, glob, os, sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1]
ev = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Stream_Id"]))
ev.sort()
# a step starts with the first launch of its build (k_build_begin; before round 5: the reset of the forest's voxel box)
first = next((k for k in ("k_build_begin", "k_bbox_reset") if any(e[2].startswith(k) for e in ev)), "k_part_hist<true, true>")
starts = [i for i, e in enumerate(ev) if e[2].startswith(first)]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3   # which step, counted from the end of the trace
a, b = starts[-back], starts[-back + 1]
t0 = prev = ev[a][0]
for s, e, n, st in ev[a:b]:
    print("%8.1f +%7.1f  gap %7.1f  stream %s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, st, n[:60]))
    prev = max(prev, e)
print("step: %.1f us" % ((ev[b][0] - t0) / 1e3))
