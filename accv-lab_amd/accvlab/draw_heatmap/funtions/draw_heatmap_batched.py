"""Same module path as the reference (…/draw_heatmap/funtions/draw_heatmap_batched.py:27-84).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in ops.py.
"""
from ..ops import draw_heatmap_batched  # noqa: F401
