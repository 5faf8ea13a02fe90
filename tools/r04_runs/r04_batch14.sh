#!/bin/bash
# A/B of the BB_ONE_READ bucket kernel (records read once, 512 threads x 8 items, coordinates through an LDS window)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
V=$R/build/variants/oneread.so
OCTREELIB_AMD_LIB=$V timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c5.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider > gpurun_out/b14_tests.log 2>&1
rc=$?; echo "variant tests rc=$rc" > gpurun_out/b14_rc.txt; tail -3 gpurun_out/b14_tests.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
  for W in headline c5shard; do
    timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline > gpurun_out/b14_${W}_base_$rep.json 2> gpurun_out/b14_${W}_base_$rep.err
    OCTREELIB_AMD_LIB=$V timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline > gpurun_out/b14_${W}_one_$rep.json 2> gpurun_out/b14_${W}_one_$rep.err
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/b14_*_*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    if "workload" in d and "ms" in d:
        kk = {k["kernel"] if isinstance(k, dict) and "kernel" in k else str(i): (k.get("avg_ms") if isinstance(k, dict) else k) for i, k in enumerate(d.get("kernels", []))} if isinstance(d.get("kernels"), list) else d.get("kernels")
        print(f.split("/")[-1], "step", round(d["ms"], 3), "build", round(d["insert_subdivide_only_ms"], 3), kk)
    else:
        sec = d.get("secondary", {})
        print(f.split("/")[-1], "step", round(d["ms_per_step"], 3), "build", sec.get("insert_subdivide_only", {}).get("ms") if isinstance(sec.get("insert_subdivide_only"), dict) else [k for k in sec if "insert" in k])
PY
