"""The exact command of the driver's multi-GPU tier - `python bench.py --gpus N` - rehearsed on the one GPU a test box
has: N ranks on device 0 (OCTL_BENCH_DEVICE), the collectives through the test-only RCCL stand-in.  What runs: the
launcher path (spawn_ranks -> torch.distributed.run), the uid broadcast, the routed insert, the topology gather and its
exit-3 checks, the exchange report, stdout_to_stderr - and the line must be ONE line under 8 KB (round 5's was 24 KB
and did not parse), with the topology / exchange blocks in the detail file.

N = 4 here: the GPU pool allows six processes on a card, and the test runner is one of them.  N = 8 itself can only run
on an 8-GPU node (the driver's tier); `tools/rehearse.sh N` is the same rehearsal from a shell."""

import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(cmd, env, out_path, err_path, conn):
    """Runs in a child of the GPU-free fork server: a process that has not touched the GPU may exec."""
    import subprocess

    with open(out_path, "w") as out, open(err_path, "w") as err:
        rc = subprocess.call(cmd, env=env, stdout=out, stderr=err, cwd=ROOT)
    conn.send(rc)


@pytest.mark.parametrize("ranks", [4])
def test_bench_gpus_n_through_the_stand_in(ranks, tmp_path):
    from tests import conftest

    mp = conftest.rank_spawner()
    if mp is None:
        pytest.skip("no fork server (it must be started before the GPU is touched: run through pytest's conftest)")
    stub = os.path.join(ROOT, "tests", "rccl_stub", "librccl_stub.so")
    if not os.path.exists(stub):
        pytest.skip("tests/rccl_stub/librccl_stub.so is not built (make stub / __graft_entry__.build())")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_rehearsal

    pts = 1_000_000
    env = dict(os.environ, OCTL_RCCL_LIBRARY=stub, OCTL_BENCH_DEVICE="0", OCTL_STUB_ARENA_MB="200",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    line, err, detail = str(tmp_path / "line.json"), str(tmp_path / "err.txt"), str(tmp_path / "detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1",
           "--points-per-rank", str(pts), "--no-cpu-baseline", "--detail", detail]
    parent, child = mp.Pipe()
    p = mp.Process(target=_run_bench, args=(cmd, env, line, err, child), daemon=True)
    p.start()
    try:
        assert parent.poll(600), "bench.py --gpus %d did not finish (a rank waiting in a collective?)" % ranks
        rc = parent.recv()
    finally:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    assert rc == 0, open(err).read()[-3000:]
    d, full = check_rehearsal.check(line, detail, ranks, "weak", pts)
    assert d["config"]["points_per_gpu"] == pts
    assert full["topology"]["rehearsal_on_one_device"]
