"""Same module path as the reference (…/batching_helpers/batched_mask_from_indices.py:20-61).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in indexing.py.
"""
from .indexing import get_mask_from_indices  # noqa: F401
from .ragged import RaggedBatch  # noqa: F401
