"""accvlab.lane_helpers — MI355X-native drop-in for the reference package of the same name
(packages/lane_helpers/accvlab/lane_helpers/__init__.py): the ``polyline`` sub-package."""
from . import polyline

__version__ = "0.1.0"
__all__ = ["__version__", "polyline"]
