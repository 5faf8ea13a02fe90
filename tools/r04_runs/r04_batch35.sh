#!/bin/bash
# k_part_scatter with non-temporal stores for the scattered records
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for rep in 1 2 3; do
  for V in base ntstore; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    for W in headline c5shard; do
      timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-secondary > gpurun_out/b35_${W}_${V}_$rep.json 2> gpurun_out/b35_${W}_${V}_$rep.err
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/b35_*_*.json")):
    d = json.load(open(f)); k = d["kernels"]
    if "workload" in d and "ms" in d:
        print(f.split("/")[-1], "build", round(d["insert_subdivide_only_ms"], 3), "part_scatter", round(k["part_scatter"]["ms_per_step"], 3))
    else:
        print(f.split("/")[-1], "step", round(d["ms_per_step"], 3), "part_scatter", round(k["part_scatter"]["ms_per_step"], 4))
PY
