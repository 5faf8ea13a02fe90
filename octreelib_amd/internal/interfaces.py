"""Mix-in for objects that carry an id (reference: internal/interfaces.py:12-32)."""

import itertools
from abc import ABC

__all__ = ["WithID"]

_fresh_ids = itertools.count()


class WithID(ABC):
    """`WithID()` draws the next number of a process-wide sequence; `WithID(n)` adopts `n`."""

    __slots__ = ("_id",)

    def __init__(self, _id=None):
        self._id = next(_fresh_ids) if _id is None else _id

    id = property(lambda self: self._id)
