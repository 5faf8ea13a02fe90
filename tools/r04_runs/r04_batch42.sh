#!/bin/bash
# one-off verification of the final tree beyond the committed tests: more seeds of the operation-sequence fuzz, of the
# bucket-path-vs-level-loop builds and of the RANSAC operator against the oracle
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 500 python tools/fuzz_more.py 16 136 > gpurun_out/b42_fuzz.txt 2>&1; echo "fuzz rc=$? $(tail -1 gpurun_out/b42_fuzz.txt)"
timeout -k 10 400 python tools/build_fuzz_more.py 2 > gpurun_out/b42_build_fuzz.txt 2>&1; echo "build fuzz rc=$? $(tail -1 gpurun_out/b42_build_fuzz.txt)"
timeout -k 10 500 python tools/ransac_stress.py 0 10 > gpurun_out/b42_ransac.txt 2>&1; echo "ransac stress rc=$? $(tail -2 gpurun_out/b42_ransac.txt | tr '\n' ' ')"
