"""Transforms on the hot path and its next rows (reference pulsarbat/transforms/__init__.py)."""

from . import transforms
from .transforms import *
from . import dedispersion
from .dedispersion import *

__all__ = transforms.__all__.copy()
__all__ += dedispersion.__all__
