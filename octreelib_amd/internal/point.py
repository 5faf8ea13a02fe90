"""Annotation aliases (reference: internal/point.py:15-16; np.float_ no longer exists in NumPy 2).
Documentation only: arrays are f64, a Point has shape (3,), a PointCloud shape (N, 3)."""

import typing

import numpy as np
from numpy.typing import NDArray

__all__ = ["Point", "PointCloud"]

_F64 = NDArray[np.float64]
Point = typing.Annotated[_F64, typing.Literal[3]]
PointCloud = typing.Annotated[_F64, typing.Literal["N", 3]]
