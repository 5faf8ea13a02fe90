#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests/test_gpu_failures.py -m gpu -q -p no:cacheprovider > gpurun_out/b7_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/b7_rc.txt
python tools/probes/pipeline_probe.py > gpurun_out/b7_pipeline.txt 2>&1; echo "pipeline rc=$?" >> gpurun_out/b7_rc.txt
python tools/probes/sparse_kernels.py > gpurun_out/b7_sparse.txt 2>&1; echo "sparse rc=$?" >> gpurun_out/b7_rc.txt
for v in base lv8 lv16 base; do
  if [ $v = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$v.so; fi
  python tools/c4_timing.py > gpurun_out/b7_c4_$v.txt 2>&1
  python bench.py --no-secondary --no-cpu-baseline --steps 5 > /dev/null 2>&1
done
unset OCTREELIB_AMD_LIB
cat gpurun_out/b7_rc.txt; tail -4 gpurun_out/b7_tests.log; tail -3 gpurun_out/b7_c4_*.txt
