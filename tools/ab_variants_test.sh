#!/bin/bash
# parity tests + short bench per library variant in build/variants
for so in build/variants/*.so; do
  name=$(basename $so .so)
  echo "== $name"
  OCTREELIB_AMD_LIB=$PWD/$so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --timeout 900 2>&1 | tail -1
done
tools/ab_variants.sh
