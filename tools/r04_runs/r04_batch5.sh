#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python tools/probes/order_probe.py > gpurun_out/b5_order.txt 2>&1; echo "order rc=$?" > gpurun_out/b5_rc.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b5_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b5_rc.txt
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b5_bench_head.json 2> gpurun_out/b5_bench.err; echo "bench rc=$?" >> gpurun_out/b5_rc.txt
python bench.py --workload c5shard --steps 3 > gpurun_out/b5_c5.json 2>/dev/null; echo "c5 rc=$?" >> gpurun_out/b5_rc.txt
OCTREELIB_AMD_LIB=$R/build/variants/ipt16.so python bench.py --workload c5shard --steps 3 > gpurun_out/b5_c5_ipt16.json 2>/dev/null; echo "c5 ipt16 rc=$?" >> gpurun_out/b5_rc.txt
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b5_bench_head2.json 2>/dev/null
cat gpurun_out/b5_rc.txt; tail -6 gpurun_out/b5_tests.log
