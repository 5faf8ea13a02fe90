#!/bin/bash
# final tree of the round: whole GPU suite, smoke, the round's profiles (default bench last), N = 2 rehearsal
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/b47_tests.log 2>&1; rc=$?; echo "tests rc=$rc" > gpurun_out/b47_rc.txt; tail -3 gpurun_out/b47_tests.log
[ $rc -eq 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/b47_smoke.txt 2>&1; echo "smoke rc=$?" >> gpurun_out/b47_rc.txt
bash tools/profile_round.sh r04 > gpurun_out/b47_profile.txt 2>&1; echo "profile rc=$?" >> gpurun_out/b47_rc.txt
bash tools/rehearse_n2.sh 2 > gpurun_out/b47_n2.txt 2>&1; echo "n2 rc=$?" >> gpurun_out/b47_rc.txt
cat gpurun_out/b47_rc.txt; tail -4 gpurun_out/b47_n2.txt
