#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b4_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/b4_rc.txt
python bench.py > gpurun_out/b4_bench.json 2> gpurun_out/b4_bench.err; echo "bench rc=$?" >> gpurun_out/b4_rc.txt
OCTREELIB_AMD_LIB=$R/build/variants/nodpp.so python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b4_bench_nodpp.json 2> gpurun_out/b4_bench_nodpp.err; echo "nodpp rc=$?" >> gpurun_out/b4_rc.txt
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/b4_bench_head2.json 2>/dev/null
OCTREELIB_AMD_LIB=$R/build/variants/rs_stamps.so python tools/rs_stamps.py > gpurun_out/b4_rs_stamps.txt 2>&1; echo "rs_stamps rc=$?" >> gpurun_out/b4_rc.txt
python tools/probes/pipeline_probe.py > gpurun_out/b4_pipeline.txt 2>&1; echo "pipeline rc=$?" >> gpurun_out/b4_rc.txt
cat gpurun_out/b4_rc.txt; tail -8 gpurun_out/b4_tests.log
