"""Same module path as the reference (…/batching_helpers/batched_indexing_ops.py:22-455).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in indexing.py.
"""
from .indexing import (  # noqa: F401
    BatchedIndexingAccess,
    BatchedInverseIndexingAccessInsert,
    BatchedInverseIndexingAccessNewTensor,
    batched_indexing_access,
    batched_indexing_write,
    batched_inverse_indexing_access,
)
from .ragged import RaggedBatch  # noqa: F401
