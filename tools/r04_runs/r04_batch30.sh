#!/bin/bash
# side stream with the highest priority (base) against default priority (noprio): bench lines + the step's timeline
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2; do
  for V in base noprio; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b30_${V}_$rep.json 2> gpurun_out/b30_${V}_$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/b30_${V}_$rep.json")); s = d["secondary"]
print("$V", $rep, "step", round(d["ms_per_step"], 3), "ransac", round(d["kernels"]["ransac"]["ms_per_step"], 3), "build", round(s["insert_subdivide_only"]["ms"], 3),
      "| sparse", round(s["sparse_scene"]["ms"], 3), round(s["sparse_scene"]["insert_subdivide_only_ms"], 3),
      "| c5", round(s["c5_shard"]["ms"], 2), round(s["c5_shard"]["insert_subdivide_only_ms"], 3),
      "| uniform", round(s["uniform_scene"]["ms"], 3), "| 2ctx", round(s["api_pipelined_2ctx"]["ms"], 3))
PY
  done
done
unset OCTREELIB_AMD_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/b30_trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline > /dev/null 2>&1
cd $R && python3 tools/probes/step_timeline.py gpurun_out/b30_trace
