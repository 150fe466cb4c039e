"""Same module path as the reference (packages/lane_helpers/accvlab/lane_helpers/polyline/functions.py:27-111).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in ops.py.
"""
from .ops import interpolate, interpolate_var_size_batch, lengths, lengths_var_size_batch  # noqa: F401
