"""accvlab.multi_tensor_copier — MI355X-native drop-in for the reference package of the same name
(public surface: packages/multi_tensor_copier/accvlab/multi_tensor_copier/__init__.py:22-28), plus the DataLoader
hook ``pack_batch`` / ``packing_collate`` / ``PackedBatch`` (extension, see packed.py)."""
from .copier import AsyncCopyHandle, release_cached_outputs, set_output_recycling, start_copy
from .packed import PackedBatch, pack_batch, packing_collate

__version__ = "0.1.0"
__all__ = ["__version__", "AsyncCopyHandle", "start_copy", "PackedBatch", "pack_batch", "packing_collate",
           "set_output_recycling", "release_cached_outputs"]
