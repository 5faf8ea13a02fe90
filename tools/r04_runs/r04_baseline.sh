#!/bin/bash
# round-4 baseline of the two largest BASELINE configs (per-kernel timers of the library)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python3 tools/c4_timing.py > gpurun_out/r04_base_c4.txt 2>&1 || exit 1
python3 bench.py --points-per-rank 125000000 --shard-of 8 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/r04_base_c5shard.json 2> gpurun_out/r04_base_c5shard.err || exit 2
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_base_bench.json 2> gpurun_out/r04_base_bench.err || exit 3
