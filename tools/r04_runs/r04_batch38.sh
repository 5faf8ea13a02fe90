#!/bin/bash
# k_compact_tiles with all of a tile's loads in flight (base) against loads behind the keep test (head)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "ransac or mask or filter or pipeline" > gpurun_out/b38_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/b38_tests.log)"
for rep in 1 2 3; do
  for V in base head; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    timeout -k 10 200 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b38_${V}_$rep.json 2> gpurun_out/b38_${V}_$rep.err
    python -c "
import json; d=json.load(open('gpurun_out/b38_${V}_$rep.json')); print('$V', $rep, round(d['ms_per_step'],3), 'apply_mask', round(d['kernels']['apply_mask']['ms_per_step'],4))"
  done
done
