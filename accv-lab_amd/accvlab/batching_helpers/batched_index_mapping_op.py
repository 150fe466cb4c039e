"""Same module path as the reference (…/batching_helpers/batched_index_mapping_op.py:22-219).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in indexing.py.
"""
from .indexing import BatchedIndexMapping, batched_index_mapping  # noqa: F401
from .ragged import RaggedBatch  # noqa: F401
