#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python tools/probes/fail_probe.py > gpurun_out/b6_fail.txt 2>&1; echo "fail probe rc=$?" > gpurun_out/b6_rc.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b6_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/b6_rc.txt
python bench.py > gpurun_out/b6_bench.json 2> gpurun_out/b6_bench.err; echo "bench rc=$?" >> gpurun_out/b6_rc.txt
python tools/probes/pipeline_probe.py > gpurun_out/b6_pipeline.txt 2>&1; echo "pipeline rc=$?" >> gpurun_out/b6_rc.txt
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b6_bench_head2.json 2>/dev/null
cat gpurun_out/b6_rc.txt; tail -6 gpurun_out/b6_tests.log
