#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b13_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/b13_rc.txt
SECONDS=0
python bench.py > gpurun_out/b13_bench.json 2> gpurun_out/b13_bench.err; echo "bench rc=$? in ${SECONDS}s" >> gpurun_out/b13_rc.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/b13_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/b13_rc.txt
cat gpurun_out/b13_rc.txt; tail -5 gpurun_out/b13_tests.log
