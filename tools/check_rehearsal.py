#!/usr/bin/env python3
"""Checks of a rehearsed N > 1 bench.py run (tools/rehearse.sh, tests/test_gpu_rehearsal.py):
   tools/check_rehearsal.py LINE_FILE DETAIL_FILE N MODE POINTS_PER_RANK
the stdout of the run is ONE line under 8 KB with the contract's keys; the topology / exchange blocks are in the
detail file and agree with it."""
import json
import sys


def check(line_path, detail_path, n, mode, pts):
    raw = open(line_path).read()
    lines = [ln for ln in raw.splitlines() if ln.strip()]
    assert len(lines) == 1, f"{len(lines)} lines on stdout"
    assert len(lines[0]) < 8192, len(lines[0])
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "rccl_ranks", "imbalance"):
        assert k in d, k
    assert "topology" not in d and "exchange" not in d and "secondary" not in d
    assert d["n_gpus"] == n and d["scaling"] == mode and d["rccl_ranks"] == n, (d["n_gpus"], d["scaling"], d["rccl_ranks"])
    full = json.load(open(detail_path))
    ex, tp = full["exchange"], full["topology"]
    assert len(ex["points_received_per_rank"]) == n and d["imbalance"] >= 1.0
    assert abs(d["imbalance"] - ex["imbalance_max_over_mean"]) < 1e-4 * d["imbalance"]
    total = sum(ex["points_received_per_rank"])
    per_rank = pts if mode == "weak" else pts // n
    assert total == per_rank * n, (total, per_rank, n)
    if mode == "weak":
        assert "strong_scaling_10M_total" in full["secondary"]
    # what the communicator and the devices say (the stand-in answers ncclCommCount / ncclCommUserRank / ncclGetVersion)
    assert tp["rccl_ranks"] == n == tp["launcher_world_size"], tp
    assert [r["rccl_user_rank"] for r in tp["ranks"]] == list(range(n)) == [r["launcher_rank"] for r in tp["ranks"]]
    assert tp["rank_order_agrees"] and tp["rehearsal_on_one_device"] and tp["distinct_devices"] == 1
    assert all(len(r["device_uuid"]) == 32 and r["pci_bus_id"] for r in tp["ranks"])
    m = tp["alltoall_bytes_rank_to_peer_last_step"]
    assert len(m) == n and all(len(row) == n for row in m)
    assert [sum(row) - row[i] for i, row in enumerate(m)] == ex["bytes_sent_to_peers_per_rank"]
    return d, full


if __name__ == "__main__":
    line_path, detail_path, n, mode, pts = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    d, full = check(line_path, detail_path, n, mode, pts)
    ex = full["exchange"]
    print(mode, d["n_gpus"], "ranks", round(d["ms_per_step"], 2), "ms/step", round(d["value"]), d["unit"],
          "| line bytes", len(open(line_path).read()), "| imbalance", round(d["imbalance"], 4),
          "| all-to-all ms", round(ex["alltoall_ms_per_step_max_over_ranks"], 3))
