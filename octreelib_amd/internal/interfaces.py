"""Mix-in for objects that carry an id (reference: internal/interfaces.py:12-32)."""

from abc import ABC
from typing import Optional

__all__ = ["WithID"]


class WithID(ABC):
    _id_static_counter = 0

    def __init__(self, _id: Optional[int] = None):
        if _id is None:
            _id = WithID._id_static_counter
            WithID._id_static_counter += 1
        self._id = _id

    @property
    def id(self):
        return self._id
