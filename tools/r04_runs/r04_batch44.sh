#!/bin/bash
# RANSAC descriptor chain: size-class starts and the position table folded into k_block_scatter (base) against
# three kernels (head); step time at 100 k, 1 M and 10 M points
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "ransac" > gpurun_out/b44_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/b44_tests.log)"
for rep in 1 2; do
  for V in base head; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    for n in 100000 1000000 10000000; do
      timeout -k 10 120 python bench.py --points $n --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V', $rep, $n, round(d['ms_per_step'],4), 'ms/step', 'prepare', round(d['kernels']['ransac_prepare']['ms_per_step'],4), 'ransac', round(d['kernels']['ransac']['ms_per_step'],4))"
    done
  done
done
