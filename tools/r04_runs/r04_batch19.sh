#!/bin/bash
# the tree with the split RANSAC launch (128 lanes x 8 hypotheses for blocks under 128 points): whole GPU suite, the
# default bench, smoke, then the round's profiles
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/b19_tests.log 2>&1; rc=$?; echo "tests rc=$rc" > gpurun_out/b19_rc.txt; tail -3 gpurun_out/b19_tests.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/b19_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/b19_rc.txt
bash tools/profile_round.sh r04 > gpurun_out/b19_profile.txt 2>&1; echo "profile rc=$?" >> gpurun_out/b19_rc.txt
cat gpurun_out/b19_rc.txt; tail -5 gpurun_out/b19_profile.txt
