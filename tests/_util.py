"""Helpers shared by the parity tests: canonical forms of leaf tables."""

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def canon(corners, edges, sizes, idx):
    """(corners (Lf,3) f64, edges (Lf,), sizes (Lf,), concatenated idx) ->
    ordered list of ((corner bytes, edge bytes), tuple(sorted idx))."""
    corners = np.ascontiguousarray(corners, dtype=np.float64).reshape(-1, 3)
    edges = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1)
    off = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    out = []
    for i in range(len(edges)):
        # +0.0 normalises a possible -0.0 corner coordinate
        key = ((corners[i] + 0.0).tobytes(), edges[i].tobytes())
        out.append((key, tuple(sorted(int(v) for v in idx[off[i] : off[i + 1]]))))
    return out


def canon_from_list(table):
    """list of (corner, edge, idx) -> same canonical form."""
    out = []
    for corner, edge, idx in table:
        key = ((np.asarray(corner, dtype=np.float64) + 0.0).tobytes(), np.float64(edge).tobytes())
        out.append((key, tuple(sorted(int(v) for v in idx))))
    return out


def golden_canon(g, prefix):
    return canon(g[f"{prefix}_corners"], g[f"{prefix}_edges"], g[f"{prefix}_sizes"], g[f"{prefix}_idx"])


def assert_same_leaves(got, want, ordered=True):
    """Exact equality of the leaf map (corner bits, edge bits) -> index set, and of the
    list order when ordered=True."""
    assert len(got) == len(want), f"{len(got)} leaves, expected {len(want)}"
    assert dict(got) == dict(want)
    if ordered:
        assert [k for k, _ in got] == [k for k, _ in want]


def set_option(name, value=1):
    """A diagnostic switch of the process-wide context (octl_debug_set_option: the library reads its OCTL_* environment
    only when a context is created).  tests/conftest.py resets every touched switch after each test."""
    from octreelib_amd import _native as nat

    nat.get_context().set_option(name, value)
