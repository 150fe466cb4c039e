"""Same module path as the reference (…/batching_helpers/data_format/set_padded_to.py:20-43).
Module path kept for code that imports from the reference's sub-modules; the implementation lives in pad_fill.py.
"""
from ..pad_fill import SetPaddedTo  # noqa: F401
