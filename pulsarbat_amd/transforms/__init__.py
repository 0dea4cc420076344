"""Transforms on the hot path (reference pulsarbat/transforms/__init__.py)."""

from . import dedispersion
from .dedispersion import *

__all__ = dedispersion.__all__.copy()
