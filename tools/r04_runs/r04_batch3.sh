#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python tools/probes/order_probe.py > gpurun_out/b3_order.txt 2>&1; echo "order rc=$?" > gpurun_out/b3_rc.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b3_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b3_rc.txt
python bench.py > gpurun_out/b3_bench.json 2> gpurun_out/b3_bench.err; echo "bench rc=$?" >> gpurun_out/b3_rc.txt
OCTREELIB_AMD_LIB=$R/build/variants/stamps.so python tools/bb_stamps.py > gpurun_out/b3_stamps.txt 2>&1; echo "stamps rc=$?" >> gpurun_out/b3_rc.txt
OCTREELIB_AMD_LIB=$R/build/variants/rs_stamps.so python tools/rs_stamps.py > gpurun_out/b3_rs_stamps.txt 2>&1; echo "rs_stamps rc=$?" >> gpurun_out/b3_rc.txt
cat gpurun_out/b3_rc.txt; tail -8 gpurun_out/b3_tests.log
