from typing import TypeVar

__all__ = ["T"]

T = TypeVar("T")
