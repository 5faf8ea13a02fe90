#!/bin/bash
# fewer launches around the host waits: tile + block kept counts in one kernel, awaited scalars written into the pinned
# mirror by the kernels themselves (base) against the commit before (head)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/b46_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/b46_tests.log)"
for rep in 1 2 3; do
  for V in base head; do
    if [ $V = base ]; then unset OCTREELIB_AMD_LIB; else export OCTREELIB_AMD_LIB=$R/build/variants/$V.so; fi
    for n in 100000 10000000; do
      timeout -k 10 120 python bench.py --points $n --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$V', $rep, $n, round(d['ms_per_step'],4), 'ms/step', 'part_hist', round(k['part_hist']['ms_per_step'],4))"
    done
  done
done
