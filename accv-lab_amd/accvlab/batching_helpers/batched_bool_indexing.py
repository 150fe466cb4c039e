"""Same module path as the reference (packages/batching_helpers/accvlab/batching_helpers/batched_bool_indexing.py).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in bool_indexing.py.
"""
from .bool_indexing import batched_bool_indexing, batched_bool_indexing_write  # noqa: F401
from .ragged import RaggedBatch  # noqa: F401
