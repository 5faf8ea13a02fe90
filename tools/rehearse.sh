#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: N ranks on device 0, the collectives through the tests'
# RCCL stand-in (shared-memory transport between the rank processes).  `python bench.py --gpus N` starts its ranks
# itself (spawn_ranks -> torch.distributed.run -> N x bench.py), exactly the command the driver's SCALE tier runs for
# N = 8.  The numbers mean nothing (the stand-in stages through host memory); what it shows is that the launcher
# contract, the uid broadcast, the routed insert, the route-ahead thread, the topology gather with its exit-3 checks,
# the exchange / imbalance report, stdout_to_stderr and the BOUNDED JSON line work with WORLD_SIZE > 1 - in both
# scaling modes (weak: the driver's line, with the fixed-total-N point under `secondary`; strong: `--scaling strong`).
# The GPU pool allows six processes on a card: N <= 5 beside a test runner, N = 8 only on a real 8-GPU node.
#   usage (through gpurun): tools/rehearse.sh [ranks] [points per rank]
N=${1:-2}
PTS=${2:-1000000}
export OCTL_RCCL_LIBRARY=$PWD/tests/rccl_stub/librccl_stub.so OCTL_BENCH_DEVICE=0 OCTL_STUB_ARENA_MB=200
mkdir -p gpurun_out
for MODE in weak strong; do
  DETAIL=gpurun_out/bench_n${N}_${MODE}_detail.json
  timeout -k 10 500 python bench.py --gpus $N --steps 3 --warmup 1 --points-per-rank $PTS --no-cpu-baseline --scaling $MODE \
    --detail $DETAIL > gpurun_out/bench_n${N}_$MODE.json 2> gpurun_out/bench_n${N}_$MODE.err
  echo "$MODE: rc=$? stdout lines: $(wc -l < gpurun_out/bench_n${N}_$MODE.json) bytes: $(wc -c < gpurun_out/bench_n${N}_$MODE.json)"
  python tools/check_rehearsal.py gpurun_out/bench_n${N}_$MODE.json $DETAIL $N $MODE $PTS || exit 1
done
