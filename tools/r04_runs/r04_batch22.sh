#!/bin/bash
# the tree after the one-read scratch-free bucket kernel, the split RANSAC launch and the chunk-path changes:
# whole GPU suite, smoke, RANSAC phase clocks, the round's profiles (default bench last), N = 2 rehearsal
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/b22_tests.log 2>&1; rc=$?; echo "tests rc=$rc" > gpurun_out/b22_rc.txt; tail -3 gpurun_out/b22_tests.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/b22_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/b22_rc.txt
OCTREELIB_AMD_LIB=$R/build/variants/rs_stamps.so timeout -k 10 300 python tools/rs_stamps.py > gpurun_out/b22_rs_stamps.txt 2>&1; echo "stamps rc=$?" >> gpurun_out/b22_rc.txt
bash tools/profile_round.sh r04 > gpurun_out/b22_profile.txt 2>&1; echo "profile rc=$?" >> gpurun_out/b22_rc.txt
bash tools/rehearse_n2.sh 2 > gpurun_out/b22_n2.txt 2>&1; echo "n2 rc=$?" >> gpurun_out/b22_rc.txt
cat gpurun_out/b22_rc.txt; cat gpurun_out/b22_rs_stamps.txt; tail -4 gpurun_out/b22_n2.txt
