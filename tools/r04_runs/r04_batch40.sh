#!/bin/bash
# k_ransac with 128 / 512 lanes per block (8 / 2 hypotheses per lane) against the shipped 256 x 4; phase clocks of the
# one-read bucket kernel
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for V in base noexit; do
  if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "ransac" -p no:cacheprovider > gpurun_out/b40_tests_$V.log 2>&1
  echo "$V tests rc=$? $(tail -1 gpurun_out/b40_tests_$V.log)"
  for rep in 1 2 3; do
    timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b40_${V}_$rep.json 2> gpurun_out/b40_${V}_$rep.err
    python -c "
import json; d=json.load(open('gpurun_out/b40_${V}_$rep.json')); print('$V', $rep, round(d['ms_per_step'],3), 'ransac', round(d['kernels']['ransac']['ms_per_step'],3))"
  done
done
