"""Same module path as the reference (…/batching_helpers/data_format/ragged_batch.py:31-1111).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in ragged.py.
"""
from ..ragged import RaggedBatch  # noqa: F401
