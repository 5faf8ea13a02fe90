#!/bin/bash
# sparse scene: kernels of the step, before / after the barrier-free chunk compaction
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider -k "bucket_build or sparse or thin" > gpurun_out/b21_tests.log 2>&1
rc=$?; echo "tests rc=$rc $(tail -1 gpurun_out/b21_tests.log)"
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
for V in head base; do
  if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
  python3 $R/tools/probes/sparse_kernels.py > $R/gpurun_out/b21_sparse_$V.txt 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b21_prof_$V -- python3 $R/tools/probes/sparse_kernels.py > /dev/null 2>&1
  echo "== $V"; cat $R/gpurun_out/b21_sparse_$V.txt
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/b21_prof_$V/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if n.startswith(("k_bucket", "k_part", "k_ransac")):
        print("   %-34s calls %4s avg %9.1f us" % (n[:34], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
