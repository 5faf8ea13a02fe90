import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("OCTREELIB_AMD_CHECKS", "1")   # the package asserts its internal invariants under the tests
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def _gpu_present() -> bool:
    return os.path.exists("/dev/kfd")


# Rank processes of the multi-rank routing test (tests/test_gpu_route_multirank.py) are forked by
# multiprocessing's fork server.  The server is started HERE - when pytest imports conftest.py, before any test
# has touched the GPU - so that no process that has initialised the GPU ever forks or execs (on the GPU pool an
# exec from such a process is refused).  The server itself never touches the GPU; its children do.
_RANK_SPAWNER = None
if _gpu_present():
    try:
        import multiprocessing as _mp
        from multiprocessing import forkserver as _forkserver

        _RANK_SPAWNER = _mp.get_context("forkserver")
        _forkserver.ensure_running()
    except Exception:  # pragma: no cover - no fork server on this platform
        _RANK_SPAWNER = None


def rank_spawner():
    """multiprocessing context whose Process objects are forked by the pre-started, GPU-free fork server."""
    return _RANK_SPAWNER


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _reset_library_options():
    """Diagnostic switches a test set on a process-wide context (tests/_util.set_option) do not outlive it."""
    yield
    nat = sys.modules.get("octreelib_amd._native")
    if nat is not None:
        for ctx in list(nat._default_ctx.values()):
            ctx.reset_options()
