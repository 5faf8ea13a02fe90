#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: two ranks on device 0, the collectives through the tests'
# RCCL stand-in (shared-memory transport between the rank processes).  The numbers mean nothing (the stand-in stages
# through host memory); what it shows is that the launcher contract, the routed insert, the route-ahead thread, the
# exchange / imbalance report and the JSON line work with WORLD_SIZE > 1 - in both scaling modes (weak: the driver's
# line, with the fixed-total-N point under `secondary`; strong: `--scaling strong`).
#   usage (through gpurun): tools/rehearse_n2.sh [ranks]
N=${1:-2}
export OCTL_RCCL_LIBRARY=$PWD/tests/rccl_stub/librccl_stub.so OCTL_BENCH_DEVICE=0 OCTL_STUB_ARENA_MB=200
PORT=29511
for MODE in weak strong; do
  PORT=$((PORT + 1))
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
    bench.py --gpus $N --steps 3 --warmup 1 --points 2000000 --no-cpu-baseline --scaling $MODE \
    > gpurun_out/bench_n${N}_$MODE.json 2> gpurun_out/bench_n${N}_$MODE.err
  echo "$MODE: rc=$? stdout lines: $(wc -l < gpurun_out/bench_n${N}_$MODE.json)"
  python - <<PY || exit 1
import json
d = json.load(open("gpurun_out/bench_n${N}_$MODE.json"))
assert d["n_gpus"] == $N and d["scaling"] == "$MODE", (d["n_gpus"], d["scaling"])
ex = d["exchange"]
assert len(ex["points_received_per_rank"]) == $N and d["imbalance"] == ex["imbalance_max_over_mean"] >= 1.0
total = sum(ex["points_received_per_rank"])
assert total == (2000000 if "$MODE" == "weak" else 2000000 // $N) * $N, total
if "$MODE" == "weak":
    assert "strong_scaling_10M_total" in d["secondary"]
# what the communicator and the devices say (the stand-in answers ncclCommCount / ncclCommUserRank / ncclGetVersion)
tp = d["topology"]
assert d["rccl_ranks"] == tp["rccl_ranks"] == $N == tp["launcher_world_size"], tp
assert [r["rccl_user_rank"] for r in tp["ranks"]] == list(range($N)) == [r["launcher_rank"] for r in tp["ranks"]]
assert tp["rank_order_agrees"] and tp["rehearsal_on_one_device"] and tp["distinct_devices"] == 1   # (every rank on device 0 here)
assert all(len(r["device_uuid"]) == 32 and r["pci_bus_id"] for r in tp["ranks"])
m = tp["alltoall_bytes_rank_to_peer_last_step"]
assert len(m) == $N and all(len(row) == $N for row in m)
assert [sum(row) - row[i] for i, row in enumerate(m)] == ex["bytes_sent_to_peers_per_rank"], (m, ex["bytes_sent_to_peers_per_rank"])
print("$MODE", d["n_gpus"], "ranks", round(d["ms_per_step"], 2), "ms/step", round(d["value"]), d["unit"],
      "| imbalance", round(d["imbalance"], 4), "| all-to-all ms", round(ex["alltoall_ms_per_step_max_over_ranks"], 3),
      "| bytes sent", ex["bytes_sent_to_peers_per_rank"], "| leaves evaluated", d["roofline_valu"]["leaves_evaluated"])
PY
done
