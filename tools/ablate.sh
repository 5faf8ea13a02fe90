#!/bin/bash
# Ablation timings of k_ransac (no scoring / no plane fit).  The switches are COMPILE-TIME (-DRS_ABLATE=1|2):
# the shipped library has none.  Builds two variants and runs the headline-only bench on each.
set -e
tools/build_variant.sh abl1 "-DRS_ABLATE=1"
tools/build_variant.sh abl2 "-DRS_ABLATE=2"
for a in abl1 abl2; do
  OCTREELIB_AMD_LIB=$PWD/build/variants/$a.so python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/$a.json 2> gpurun_out/$a.err || echo FAILED $a
done
