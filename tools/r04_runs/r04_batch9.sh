#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "prefix or six_digits or single_cube" > gpurun_out/b9_tests_a.log 2>&1; echo "tests a rc=$?" > gpurun_out/b9_rc.txt
tail -15 gpurun_out/b9_tests_a.log
python bench.py --workload c4 --steps 4 > gpurun_out/b9_c4.json 2> gpurun_out/b9_c4.err; echo "c4 rc=$?" >> gpurun_out/b9_rc.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b9_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b9_rc.txt
cat gpurun_out/b9_rc.txt; tail -6 gpurun_out/b9_tests.log
