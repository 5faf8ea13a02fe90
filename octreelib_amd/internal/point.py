"""Annotation aliases (reference: internal/point.py:15-16; np.float_ no longer exists in NumPy 2)."""

from typing import Annotated, Literal

import numpy as np
import numpy.typing as npt

__all__ = ["Point", "PointCloud"]

Point = Annotated[npt.NDArray[np.float64], Literal[3]]
PointCloud = Annotated[npt.NDArray[np.float64], Literal["N", 3]]
