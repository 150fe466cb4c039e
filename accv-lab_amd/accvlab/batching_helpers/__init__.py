"""accvlab.batching_helpers — MI355X-native drop-in for the reference package of the same name
(public surface: packages/batching_helpers/accvlab/batching_helpers/__init__.py:23-66).

``RaggedBatch`` plus fifteen functions; GPU tensors run hand-written gfx950 kernels from libaccv_hip.so
(ragged gather / scatter / pair mapping / constant insert / pad fill / ballot compaction / pack), CPU
tensors use torch.  The reference's private extension modules are available under their original names
(``batched_indexing_access_cuda``, ``batched_indexing_access_cpu``) with the same functions.
"""
# Imported through the modules that carry the reference's names (thin aliases of ragged / indexing / bool_indexing /
# packing), as the reference's own __init__ does: a module and a function share the name `batched_bool_indexing`, and
# the function must be what the package attribute ends up bound to.
from .data_format import RaggedBatch
from .batched_indexing_ops import (
    batched_indexing_access,
    batched_inverse_indexing_access,
    batched_indexing_write,
)
from .batched_index_mapping_op import batched_index_mapping
from .batched_mask_from_indices import get_mask_from_indices
from .batched_bool_indexing import (
    batched_bool_indexing,
    batched_bool_indexing_write,
)
from .batched_processing_py import (
    get_compact_from_named_tuple,
    get_compact_lists,
    get_indices_from_mask,
    average_over_targets,
    sum_over_targets,
    apply_mask_to_tensor,
    squeeze_except_batch_and_sample,
    combine_data,
)

from .fused import matched_pair_loss_sum  # (extension) gathers + per-object loss + masked sum in one launch

__version__ = "0.1.0"

__all__ = [
    "__version__",
    "RaggedBatch",
    "apply_mask_to_tensor",
    "average_over_targets",
    "batched_bool_indexing",
    "batched_bool_indexing_write",
    "batched_index_mapping",
    "batched_indexing_access",
    "batched_indexing_write",
    "batched_inverse_indexing_access",
    "combine_data",
    "get_compact_from_named_tuple",
    "get_compact_lists",
    "get_indices_from_mask",
    "get_mask_from_indices",
    "matched_pair_loss_sum",
    "squeeze_except_batch_and_sample",
    "sum_over_targets",
]
