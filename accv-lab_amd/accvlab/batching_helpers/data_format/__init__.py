"""Same sub-package path as the reference (…/batching_helpers/data_format/__init__.py:15-17).
"""
from ..ragged import RaggedBatch  # noqa: F401

__all__ = []
