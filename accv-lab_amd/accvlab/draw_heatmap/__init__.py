"""accvlab.draw_heatmap — MI355X-native drop-in for the reference package of the same name.

Public API (same names, argument meaning and error behaviour as
packages/draw_heatmap/accvlab/draw_heatmap/__init__.py:22-24 of the reference):
``draw_heatmap`` (flat / "concatenated" input) and ``draw_heatmap_batched`` (RaggedBatch input, optional
class-wise planes).  Both accept one extra keyword, ``clear=False``: with ``clear=True`` the map is
overwritten with max(0, splats) in a single write-only pass (fused zero-fill + draw).
Extensions next to them: ``get_centers_and_radii`` (bbox -> centre/radius front end) and ``draw_polylines_batched`` /
``sample_lane_targets`` (lane raster = polyline sampler + splat), ``draw_heatmap_multiscale`` (all strides of a batch in one
launch), ``draw_targets_multiscale`` (box maps + lane maps of a step in two launches).
"""
from .lanes import (draw_polylines_batched, draw_polylines_multiscale, draw_targets_multiscale, sample_lane_targets,
                    sample_lanes)
from .ops import draw_heatmap, draw_heatmap_batched, draw_heatmap_multiscale, get_centers_and_radii

__version__ = "0.1.0"
__all__ = ["__version__", "draw_heatmap", "draw_heatmap_batched", "get_centers_and_radii", "draw_polylines_batched",
           "draw_heatmap_multiscale", "draw_polylines_multiscale", "draw_targets_multiscale", "sample_lane_targets", "sample_lanes"]
