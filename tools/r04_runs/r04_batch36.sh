#!/bin/bash
# k_part_scatter storing whole sectors (neighbouring lanes share a record's two halves) against two stores per lane
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
OCTREELIB_AMD_LIB=$R/build/variants/pair.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bucket or golden or six_digits or prefix or hint" > gpurun_out/b36_tests.log 2>&1
echo "pair tests rc=$? $(tail -1 gpurun_out/b36_tests.log)"
for rep in 1 2 3; do
  for V in base pair; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    for W in headline c5shard; do
      timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-secondary > gpurun_out/b36_${W}_${V}_$rep.json 2> gpurun_out/b36_${W}_${V}_$rep.err
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/b36_*_*.json")):
    d = json.load(open(f)); k = d["kernels"]
    sc = k.get("part_scatter", k.get("prefix_scatter"))
    if "workload" in d and "ms" in d:
        print(f.split("/")[-1], "step", round(d["ms"], 2), "build", round(d["insert_subdivide_only_ms"], 3), "scatter", round(sc["ms_per_step"], 3))
    elif "workload" in d:
        print(f.split("/")[-1], {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in d.items() if "ms" in kk}, "scatter", round(sc["ms_per_step"], 3))
    else:
        print(f.split("/")[-1], "step", round(d["ms_per_step"], 3), "scatter", round(sc["ms_per_step"], 4))
PY
