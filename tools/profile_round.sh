#!/bin/bash
# Collect the profiles kept under profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh TAG      e.g. TAG=r03
# Every rocprofv3 pass profiles the HEADLINE scene only (bench.py --no-secondary --no-cpu-baseline: the timed
# region of BASELINE config 3 and nothing else), so that the per-kernel averages of the tracked files are the
# headline's own - the secondaries (uniform scene, two streams, Python API ...) launch the same kernels on other
# inputs and used to be averaged in.
# 1. (last, see 6.) full default bench.py run                   -> profiles/TAG_bench.json
# 2. rocprofv3 --kernel-trace --stats of the headline-only run    -> profiles/TAG_kernel_stats.csv
# 3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters), as MI355X_MICROARCH.md asks
# 4. the same kernel-stats pass on the uniform scene               -> profiles/TAG_uniform_kernel_stats.csv
# then tools/summarize_profiles.py condenses them into profiles/TAG_*.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
HEAD="--no-secondary --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 $HEAD > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_$TAG.err || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 $HEAD > /dev/null 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 $HEAD > /dev/null 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_sq_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 $HEAD > /dev/null 2>&1 || exit 5
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 --output-format csv -d $R/gpurun_out/pmc_valu_$TAG -- python3 $R/bench.py --steps 3 --warmup 1 $HEAD > /dev/null 2>&1 || exit 5
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_uniform -- python3 $R/bench.py --steps 10 --warmup 2 --cloud uniform $HEAD > $R/gpurun_out/prof_${TAG}_uniform_bench.json 2> $R/gpurun_out/prof_${TAG}_uniform.err || exit 6
cd $R && python3 tools/summarize_profiles.py $TAG gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG gpurun_out/prof_${TAG}_bench.json "bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline: the headline scene only, 10 M planar points, three resident clouds in turn" gpurun_out/pmc_valu_$TAG || exit 6
python3 tools/probes/step_timeline.py gpurun_out/prof_$TAG planar 4 > profiles/${TAG}_step_timeline.txt 2>&1
# what k_ransac executes of what the algorithm asks for: the counting variant of the library (tools/build_variant.sh)
if [ -f build/variants/rs_counts.so ]; then
  OCTREELIB_AMD_LIB=$R/build/variants/rs_counts.so python3 tools/rs_counts.py > gpurun_out/rs_counts.log 2>&1 && cp gpurun_out/rs_counts.json profiles/${TAG}_ransac_counts.json
fi
# the opcode ledger of the benchmarked k_ransac instance: static histogram by phase x the execution counts above
python3 tools/ransac_isa.py profiles/${TAG}_ransac_counts.json > gpurun_out/ransac_isa.txt 2> gpurun_out/ransac_isa.err && [ -s gpurun_out/ransac_isa.txt ] && cp gpurun_out/ransac_isa.txt profiles/${TAG}_ransac_isa.txt || echo "profile_round: opcode ledger FAILED (gpurun_out/ransac_isa.err) - profiles/${TAG}_ransac_isa.txt left as it was" >&2
# one step of a dense 100 k-point scan, kernel by kernel
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_small -- python3 $R/bench.py --workload small --steps 20 --warmup 3 > /dev/null 2>&1)
python3 tools/probes/step_timeline.py gpurun_out/prof_${TAG}_small planar 40 > profiles/${TAG}_small_scan_timeline.txt 2>&1
cp $(ls gpurun_out/prof_${TAG}_uniform/*/*kernel_stats.csv | head -1) profiles/${TAG}_uniform_kernel_stats.csv
# 5. the two largest BASELINE configs alone (bench.py --workload c4 / c5shard): kernel stats + FETCH / WRITE passes
#    -> profiles/TAG_c4_*, profiles/TAG_c5shard_*
if [ -z "$OCTL_PROFILE_HEADLINE_ONLY" ]; then
  cd /tmp
  for W in c4 c5shard; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$W -- python3 $R/bench.py --workload $W --steps 3 > $R/gpurun_out/prof_${TAG}_${W}_bench.json 2> $R/gpurun_out/prof_${TAG}_$W.err || exit 7
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch_${TAG}_$W -- python3 $R/bench.py --workload $W --steps 3 > /dev/null 2>&1 || exit 8
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write_${TAG}_$W -- python3 $R/bench.py --workload $W --steps 3 > /dev/null 2>&1 || exit 9
    (cd $R && python3 tools/summarize_profiles.py ${TAG}_$W gpurun_out/prof_${TAG}_$W gpurun_out/pmc_fetch_${TAG}_$W gpurun_out/pmc_write_${TAG}_$W - gpurun_out/prof_${TAG}_${W}_bench.json "bench.py --workload $W --steps 3") || exit 10
  done
  cd $R
fi
# 6. the full default bench.py run LAST: its counter figures (roofline.traffic, roofline_build) are read from the
#    profiles/TAG_* files the passes above have just written
cd /tmp && python3 $R/bench.py --detail $R/gpurun_out/${TAG}_bench_detail.json > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || exit 1
cd $R && cp gpurun_out/${TAG}_bench.json profiles/${TAG}_bench.json && cp gpurun_out/${TAG}_bench_detail.json profiles/${TAG}_bench_detail.json
# profiles/ is not writable back from the box: the condensed files are copied to gpurun_out/ too
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
