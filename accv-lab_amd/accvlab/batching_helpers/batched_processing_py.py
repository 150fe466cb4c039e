"""Same module path as the reference (…/batching_helpers/batched_processing_py.py:23-628).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in packing.py and
bool_indexing.py.
"""
from .bool_indexing import (  # noqa: F401
    batched_bool_indexing,
    get_compact_from_named_tuple,
    get_compact_lists,
    get_indices_from_mask,
)
from .packing import (  # noqa: F401
    apply_mask_to_tensor,
    average_over_targets,
    combine_data,
    squeeze_except_batch_and_sample,
    sum_over_targets,
)
from .ragged import RaggedBatch  # noqa: F401
