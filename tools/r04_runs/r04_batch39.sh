#!/bin/bash
# histogram pass with 512 / 256 threads per supertile instead of 1024 (768 supertiles: 1.5 rounds of 2 resident workgroups)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for V in ph512 ph256; do
OCTREELIB_AMD_LIB=$R/build/variants/$V.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bucket or golden or hint or prefix" > gpurun_out/b39_tests_$V.log 2>&1
echo "$V tests rc=$? $(tail -1 gpurun_out/b39_tests_$V.log)"
done
for rep in 1 2 3; do
  for V in base ph512 ph256; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    for W in headline c5shard; do
      timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-secondary > gpurun_out/b39_${W}_${V}_$rep.json 2> gpurun_out/b39_${W}_${V}_$rep.err
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/b39_*_*.json")):
    d = json.load(open(f)); k = d["kernels"]
    if "workload" in d and "ms" in d:
        print(f.split("/")[-1], "build", round(d["insert_subdivide_only_ms"], 3), "part_hist", round(k["part_hist"]["ms_per_step"], 3))
    else:
        print(f.split("/")[-1], "step", round(d["ms_per_step"], 3), "part_hist", round(k["part_hist"]["ms_per_step"], 4))
PY
