#!/bin/bash
# one GPU call: whole GPU suite (no -x), default bench line, phase clocks of the build kernels
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b2_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/b2_rc.txt
python bench.py > gpurun_out/b2_bench.json 2> gpurun_out/b2_bench.err; echo "bench rc=$?" >> gpurun_out/b2_rc.txt
OCTREELIB_AMD_LIB=$R/build/variants/stamps.so python tools/bb_stamps.py > gpurun_out/b2_stamps.txt 2>&1; echo "stamps rc=$?" >> gpurun_out/b2_rc.txt
cat gpurun_out/b2_rc.txt; tail -5 gpurun_out/b2_tests.log
