#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/b8_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/b8_rc.txt
tail -4 gpurun_out/b8_tests.log
bash tools/profile_round.sh r04 > gpurun_out/b8_profile.log 2>&1; echo "profile rc=$?" >> gpurun_out/b8_rc.txt
cat gpurun_out/b8_rc.txt; tail -5 gpurun_out/b8_profile.log
