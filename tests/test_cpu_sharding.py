"""N > 1 path on CPU: world_size 2 and 3 over gloo (partition by owner hash, counts exchange,
all-to-all, local build, global counters)."""

import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_build_matches_single_process(world):
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(ROOT, "tests", "_dist_worker.py"),
    ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "SHARDING_OK" in out.stdout
