"""Same sub-package path (and spelling) as the reference (packages/draw_heatmap/accvlab/draw_heatmap/funtions/__init__.py:26-30).
"""
from ..ops import draw_heatmap  # noqa: F401
from .draw_heatmap_batched import draw_heatmap_batched  # noqa: F401  (module and function share the name)

__all__ = ["draw_heatmap", "draw_heatmap_batched"]
