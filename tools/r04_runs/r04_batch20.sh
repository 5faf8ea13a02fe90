#!/bin/bash
# bucket kernel without scratch (unroll 1 on the per-thread bin loops, 7 of 8 rounds of coordinates in registers)
# against the committed one (132 B of scratch per lane)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py -m gpu -q -x -p no:cacheprovider > gpurun_out/b20_tests.log 2>&1
rc=$?; echo "tests rc=$rc $(tail -1 gpurun_out/b20_tests.log)"
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
  for V in prev base keep6 batch4; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    for W in headline c5shard; do
      timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-secondary > gpurun_out/b20_${W}_${V}_$rep.json 2> gpurun_out/b20_${W}_${V}_$rep.err
    done
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/b20_*_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    k = d["kernels"]
    if "workload" in d and "ms" in d:
        print(f.split("/")[-1], "step", round(d["ms"], 3), "build", round(d["insert_subdivide_only_ms"], 3), "bucket_build", round(k["bucket_build"]["ms_per_step"], 3))
    else:
        print(f.split("/")[-1], "step", round(d["ms_per_step"], 3), "bucket_build", round(k["bucket_build"]["ms_per_step"], 4), "ransac", round(k["ransac"]["ms_per_step"], 3))
PY
