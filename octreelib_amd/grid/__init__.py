"""Grid, GridConfig and their abstract bases."""

from octreelib_amd.grid.grid_base import GridBase, GridConfigBase, GridVisualizationType, VisualizationConfig
from octreelib_amd.grid.grid import Grid, GridConfig

__all__ = ["GridVisualizationType", "VisualizationConfig", "GridConfigBase", "GridBase", "Grid", "GridConfig"]
