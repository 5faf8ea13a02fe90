#!/bin/bash
# Collect the profiles kept under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh TAG      e.g. TAG=r01
# 1. full default bench.py run                      -> gpurun_out/TAG_bench.json
# 2. rocprofv3 --kernel-trace --stats of bench.py   -> gpurun_out/prof_TAG
# 3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters), as MI355X_MICROARCH.md asks
# then tools/summarize_profiles.py condenses them into profiles/TAG_*.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 5
cd $R && python3 tools/summarize_profiles.py $TAG gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG gpurun_out/prof_${TAG}_bench.json && cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json
# profiles/ is not writable back from the box: the condensed files are copied to gpurun_out/ too
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
