#!/bin/bash
# one GPU call: whole GPU suite (no -x), default bench line, 2-rank rehearsal
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b1_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/b1_rc.txt
python bench.py > gpurun_out/b1_bench.json 2> gpurun_out/b1_bench.err; echo "bench rc=$?" >> gpurun_out/b1_rc.txt
bash tools/rehearse_n2.sh 2 > gpurun_out/b1_rehearse.log 2>&1; echo "rehearse rc=$?" >> gpurun_out/b1_rc.txt
cat gpurun_out/b1_rc.txt; tail -5 gpurun_out/b1_tests.log
