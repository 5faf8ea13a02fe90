#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: two ranks on device 0, the collectives through the tests'
# RCCL stand-in (shared-memory transport between the rank processes).  The numbers mean nothing (the stand-in stages
# through host memory); what it shows is that the launcher contract, the routed insert, the route-ahead thread and
# the JSON line work with WORLD_SIZE > 1.   usage (through gpurun): tools/rehearse_n2.sh [ranks]
N=${1:-2}
export OCTL_RCCL_LIBRARY=$PWD/tests/rccl_stub/librccl_stub.so OCTL_BENCH_DEVICE=0 OCTL_STUB_ARENA_MB=200
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus $N --steps 3 --warmup 1 --points 2000000 --no-cpu-baseline > gpurun_out/bench_n$N.json 2> gpurun_out/bench_n$N.err
echo "rc=$? stdout lines: $(wc -l < gpurun_out/bench_n$N.json)"
python - <<PY
import json
d = json.load(open("gpurun_out/bench_n$N.json"))
print(d["n_gpus"], "ranks", d["ms_per_step"], "ms/step", d["value"], d["unit"], "| leaves evaluated", d["roofline_valu"]["leaves_evaluated"])
PY
