#!/bin/bash
# host-wait latency: HSA signal waits by interrupt (default) or by polling (HSA_ENABLE_INTERRUPT=0)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2 3; do
  for V in irq poll; do
    if [ $V = irq ]; then unset HSA_ENABLE_INTERRUPT; else export HSA_ENABLE_INTERRUPT=0; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/b33_${V}_$rep.json 2> gpurun_out/b33_${V}_$rep.err
    python - <<PY
import json
d = json.load(open("gpurun_out/b33_${V}_$rep.json")); s = d["secondary"]
print("$V", $rep, "step", round(d["ms_per_step"], 3), "ransac", round(d["kernels"]["ransac"]["ms_per_step"], 3), "build", round(s["insert_subdivide_only"]["ms"], 3),
      "| sparse", round(s["sparse_scene"]["ms"], 3), "| c5", round(s["c5_shard"]["ms"], 2),
      "| uniform", round(s["uniform_scene"]["ms"], 3), "| 2ctx", round(s["api_pipelined_2ctx"]["ms"], 3), "| api_incl", round(s["api_inclusive"]["ms"], 3), "| pcie_pipe", round(s["pcie_pipelined"]["ms"], 3), "| late", str(s["late_poses"])[:80])
PY
  done
done
