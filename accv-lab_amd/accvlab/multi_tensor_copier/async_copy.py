"""Same module path as the reference (packages/multi_tensor_copier/accvlab/multi_tensor_copier/async_copy.py:27-169).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in copier.py.
"""
from .copier import AsyncCopyHandle, start_copy  # noqa: F401
